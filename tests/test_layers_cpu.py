"""Host-side pieces of the layer mirrors that need no GPU."""
import pytest
import torch


def test_get_act_fn_names_and_error(pkg):
    """vllm/model_executor/layers/activation.py:8-21: four names, case-insensitive, tanh
    approximation for both gelu_new and gelu_fast, ValueError for anything else."""
    from vllm_amd.layers import get_act_fn
    x = torch.linspace(-4.0, 4.0, 101)
    assert torch.equal(get_act_fn("gelu")(x), torch.nn.functional.gelu(x))
    for name in ("gelu_new", "GELU_FAST"):
        assert torch.equal(get_act_fn(name)(x), torch.nn.functional.gelu(x, approximate="tanh"))
    assert torch.equal(get_act_fn("ReLU")(x), torch.relu(x))
    with pytest.raises(ValueError):
        get_act_fn("swish")
