class FwOp:
    """Operator tag the caller passes back as `op=` (attention.py:44,66); the
    MI355X kernel is the only implementation, so it carries no state."""
    NAME = "mi355x_prompt_attention"
