"""The entry point's smoke() (one small decode step checked against the CPU oracle) as part of the GPU suite."""
import pytest


@pytest.mark.gpu
def test_entry_point_smoke_step_agrees_with_the_oracle(capsys):
    import __graft_entry__
    __graft_entry__.smoke()
    assert "smoke ok" in capsys.readouterr().out
