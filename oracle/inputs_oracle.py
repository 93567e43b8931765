"""Loop-for-loop restatement of Worker._prepare_inputs
(vllm/worker/worker.py:128-237) on plain Python lists.  TEST INFRASTRUCTURE:
the checker for vllm-v0.1.0_amd/worker/inputs.py; small cases only."""
from typing import Any, Dict, List, Sequence


def prepare_inputs_lists(seq_group_metadata_list: Sequence[Any], block_size: int) -> Dict[str, Any]:
    input_tokens: List[int] = []
    input_positions: List[int] = []
    slot_mapping: List[int] = []
    prompt_lens: List[int] = []
    seq_groups = []
    for g in seq_group_metadata_list:                      # :139-172
        if not g.is_prompt:
            continue
        seq_ids = list(g.seq_data.keys())
        seq_groups.append((seq_ids, g.sampling_params))
        seq_id = seq_ids[0]
        prompt_tokens = g.seq_data[seq_id].get_token_ids()
        prompt_len = len(prompt_tokens)
        prompt_lens.append(prompt_len)
        input_tokens.extend(prompt_tokens)
        input_positions.extend(range(len(prompt_tokens)))
        if g.block_tables is None:
            slot_mapping.extend([0] * prompt_len)
            continue
        block_table = g.block_tables[seq_id]
        for i in range(prompt_len):
            slot_mapping.append(block_table[i // block_size] * block_size + i % block_size)
    max_context_len = 0                                      # :175-207
    max_num_blocks_per_seq = 0
    context_lens: List[int] = []
    generation_block_tables: List[List[int]] = []
    for g in seq_group_metadata_list:
        if g.is_prompt:
            continue
        seq_ids = list(g.seq_data.keys())
        seq_groups.append((seq_ids, g.sampling_params))
        for seq_id in seq_ids:
            seq_data = g.seq_data[seq_id]
            input_tokens.append(seq_data.get_last_token_id())
            context_len = seq_data.get_len()
            position = context_len - 1
            input_positions.append(position)
            block_table = g.block_tables[seq_id]
            generation_block_tables.append(block_table)
            max_context_len = max(max_context_len, context_len)
            max_num_blocks_per_seq = max(max_num_blocks_per_seq, len(block_table))
            context_lens.append(context_len)
            slot_mapping.append(block_table[position // block_size] * block_size + position % block_size)
    pad = lambda x: x + [0] * ((-len(x)) % 8)                # :209-212, :304-305
    padded_tables = [list(t) + [0] * (max_num_blocks_per_seq - len(t)) for t in generation_block_tables]
    return {
        "input_tokens": pad(input_tokens), "input_positions": pad(input_positions),
        "slot_mapping": slot_mapping, "context_lens": context_lens, "block_tables": padded_tables,
        "prompt_lens": prompt_lens, "max_context_len": max_context_len, "seq_groups": seq_groups,
    }
