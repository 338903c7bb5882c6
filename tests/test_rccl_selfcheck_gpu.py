"""The N > 1 code path on the hardware this suite gets (ONE GPU): a one-rank nccl (= RCCL) process group under
torch.distributed.run drives bench.py's per-step all_gather_into_tensor of the device-resident records, its barrier and
max-over-ranks timing, and — in the search leg — the product's VSM._score_sharded device path (records written by the engine
into HBM, gathered, ONE D2H copy).  What the driver's `--gpus 8` launch adds to this is ranks, not code."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_under_torchrun_one_rank_nccl(cuda):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--tiny",
           "--batch", "4", "--search-targets", "3", "--no-cpu-baseline", "--rccl-selfcheck", "--stream-samples", "8"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert lines[-1].startswith("{"), lines[-3:]       # the JSON line is the LAST line (RCCL's banner must not trail it)
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 1 and line["collective"]["backend"].startswith("nccl") and line["collective"]["ranks"] == 1
    assert line["value"] > 0
    for leg in ("search", "search_grouped"):
        assert "error" not in line[leg], line[leg]
        assert line[leg]["crops_scored"] == 3 * 21
    # the best-first stream leg through the same one-rank RCCL group: crop sharding with the per-step record all-gather
    st = line["search_stream"]
    assert "error" not in st, st
    assert st["shard"] == "crops" and st["ranks"] == 1 and st["searches"] == 8 and st["useful_crops"] >= 8
    assert st["stage_s"]["record_allgather_and_d2h"] > 0


def test_device_record_gather_equals_host_records(cuda):
    """VSM._score_sharded with an initialised nccl group (one rank): records gathered on the device == the host-record path."""
    import torch.distributed as dist
    from vstar_amd import preprocess as pp
    from vstar_amd.config import VSMConfig
    from vstar_amd.engine import VstarEngine
    from vstar_amd.synthetic import synthetic_image
    from vstar_amd.vsm import VSM
    from vstar_amd.weights import template_chain, trained_like_state_dict
    cfg = VSMConfig.tiny(max_batch=4, max_text_len=96)
    eng = VstarEngine(cfg, 0)
    tok = pp.SyntheticTokenizer(cfg.llm_vocab)
    eng.load_state_dict(trained_like_state_dict(cfg, seed=5, dtype=torch.bfloat16, share_layers=False, chain=template_chain(tok)))
    vsm = VSM(None, engine=eng, tokenizer=tok, strict_template=True)
    img = synthetic_image(900, 600, 4)
    vsm.set_image(img)
    boxes = [[0, 0, 900, 600], [0, 0, 450, 300], [450, 0, 450, 300], [0, 300, 450, 300], [450, 300, 450, 300]]
    q = pp.LOCATE_QUESTION.format("kite")
    host = vsm.inference_boxes(boxes, q, mode="detection", upsample=False)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        devp = vsm.inference_boxes(boxes, q, mode="detection", upsample=False)
        assert vsm.timers["gather_s"] > 0
        # the C-ABI's own collective (SURVEY §8b `vstar_allgather_results`): the engine gets an RCCL communicator over the group's
        # ranks (id broadcast through the torch group, ncclCommInitRank, ncclAllGather on the engine's stream) and the product path
        # switches to it
        from vstar_amd.dist import engine_comm_init
        engine_comm_init(eng)
        assert eng.comm_world == 1
        local = torch.arange(3 * eng.lib_result_floats(), dtype=torch.float32, device="cuda:0").reshape(3, -1)
        got = eng.allgather_results(local)
        assert torch.equal(got, local)
        devc = vsm.inference_boxes(boxes, q, mode="detection", upsample=False)
    finally:
        dist.destroy_process_group()
    for a, b, c in zip(host, devp, devc):
        assert all(torch.equal(x, y) for x, y in zip(a, b))
        assert all(torch.equal(x, y) for x, y in zip(a, c))
    eng.close()


def test_comm_entry_points_report_misuse(cuda):
    from vstar_amd._lib import VstarError
    from vstar_amd.config import VSMConfig
    from vstar_amd.engine import VstarEngine
    eng = VstarEngine(VSMConfig.tiny(max_batch=2), 0)
    with pytest.raises(VstarError, match="vstar_comm_init"):
        eng.comm_world = 1
        eng.allgather_results(torch.zeros(1, eng.lib_result_floats(), device="cuda:0"))
    eng.close()
