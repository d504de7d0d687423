"""Engine pipeline stages (bigdl_amd/parallel/pipeline.py) against the unsplit engine: two and three stage processes
share the one GPU of the box, hidden states and tokens cross over gloo."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("stages", [2, 3])
def test_engine_stages_match_the_whole_engine(tmp_path, stages):
    port = _port()
    out = str(tmp_path / "res")
    procs = []
    for r in range(stages):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(stages), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PYTHONPATH=os.path.dirname(HERE))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "pp_engine_worker.py"), out], env=env))
    codes = [p.wait(timeout=240) for p in procs]
    assert codes == [0] * stages
    res = [json.load(open(f"{out}.{r}")) for r in range(stages)]
    whole = res[0]["whole"]
    for r in range(stages):
        assert res[r]["pipe"] == whole, f"stage {r}: pipelined tokens differ from the unsplit engine"
        assert res[r]["pipe_mb"] == res[0]["whole_mb"], f"stage {r}: micro-batched pipeline differs from the unsplit engine"
        # every stage ran every (step, micro-batch) pair: prompt + 11 decode steps for each of the micro-batches
        assert res[r]["ticks"] == 12 * stages
    eos = res[0]["eos"]
    stopped = res[0]["stopped"][0][37:]
    first_hit = whole[0][37:].index(eos)
    assert stopped[: first_hit + 1] == whole[0][37: 37 + first_hit + 1] and stopped[-1] == eos
    assert all(res[r]["stopped"] == res[0]["stopped"] for r in range(stages))
