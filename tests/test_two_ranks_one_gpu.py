"""Two ranks of the REAL engine on the one GPU a test box has (round 4): the data-parallel search of BASELINE configs 3 / 4
— process group, crop sharding with the per-step record all-gather, sample sharding, the stream driver with its prefetch threads and
asynchronous uploads, the adaptive speculation priors — run as two processes under torch.distributed.run, over gloo because RCCL
refuses two ranks on one device (the records then travel as host tensors: VSM._score_sharded's non-nccl branch).  Both shard modes
must print exactly what the single-process run prints.  What it cannot show is RCCL with N ranks; what it does show is that N
processes driving real engines form identical batches step after step and agree on every decision."""
import json
import os
import subprocess
import sys

import pytest

from test_host import _free_port, _make_bench_folder

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("shard", ["crops", "samples"])
def test_two_real_engine_ranks_on_one_gpu_equal_the_single_process_run(cuda, tmp_path, shard):
    folder = str(tmp_path / "bench")
    _make_bench_folder(folder)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--benchmark-folder", folder, "--vsm-factory", "_real_tiny_vsm:make", "--confidence_high", "2.0", "--confidence_low", "0.0",
              "--target_cue_threshold", "-1", "--target_cue_threshold_minimum", "-1", "--minimum_size", "160"]
    one_json, two_json = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "visual_search.py"), *common, "--output_path", one_json],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert one.returncode == 0, one.stderr[-3000:]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "visual_search.py"), *common, "--shard", shard, "--output_path", two_json]
    two = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=dict(env, VSTAR_DIST_BACKEND="gloo", VSTAR_DIST_DEVICE="0"))
    assert two.returncode == 0, two.stderr[-3000:]
    metric_lines = lambda text: [l for l in text.splitlines() if l.startswith(("Avg search path length", "Top 1 Acc"))]  # noqa: E731
    assert len(metric_lines(two.stdout)) == 2 and metric_lines(two.stdout) == metric_lines(one.stdout)
    a, b = json.load(open(one_json)), json.load(open(two_json))
    assert b["world_size"] == 2 and b["shard"] == shard
    assert a["hits"] == b["hits"] and a["path_lengths"] == b["path_lengths"] and len(a["hits"]) == 4
    sa, sb = a["rank0_search_stats"], b["rank0_search_stats"]
    assert sa["useful_crops"] == 4 * 21                      # exhaustive depth-3 trees: the searches really descended
    if shard == "crops":                                     # every rank walks every search; each step's crops are dealt over the ranks
        # (the NUMBER of engine steps may differ: one process starts before its whole first window is loaded, sharded ranks do not)
        assert sb["useful_crops"] == sa["useful_crops"] and sb["engine_steps"] >= 21
