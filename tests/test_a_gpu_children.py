"""Multi-process entry points proven on ONE GPU: bench.py and tools/sample_sharded.py started
the way the driver starts them for N > 1 -- `python -m torch.distributed.run --nproc-per-node 1 ...`, RCCL process
group initialised in the child, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.

The file sorts FIRST in tests/ on purpose: the children are started from a pytest process that has not touched the
GPU yet (this pool refuses an exec from a process that has initialised the GPU; `torch.cuda.device_count()` does not
initialise it).  If the GPU is already initialised in this process, the tests skip instead of forking."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(script_args, timeout=600):
    if torch.cuda.device_count() < 1:
        pytest.skip("needs a GPU")
    if torch.cuda.is_initialized():
        pytest.skip("this process has initialised the GPU already: child programs are started before any GPU call only")
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("RANK", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port())] + script_args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, f"{' '.join(cmd)}\n--- stdout\n{out.stdout[-3000:]}\n--- stderr\n{out.stderr[-3000:]}"
    return out.stdout


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_bench_train_under_torchrun_one_rank():
    """the training workload exactly as the driver launches it for N > 1 (one rank here): RCCL group up, the flat
    gradient all-reduced over it, ONE JSON line with the contract's fields"""
    rec = _json_line(_torchrun(["bench.py", "--gpus", "1", "--workload", "train", "--steps", "2", "--warmup", "1"]))
    assert rec["n_gpus"] == 1 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["unit"] == "graphs/s"
    assert rec["scaling"] == "weak" and rec["higher_is_better"] is True and rec["value"] > 0
    assert rec["roofline"]["bound"] == "mfma" and 0 < rec["roofline"]["frac"] < 1
    assert rec["final_loss"] == rec["final_loss"]  # not NaN


def test_bench_sampling_under_torchrun_one_rank():
    rec = _json_line(_torchrun(["bench.py", "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                                "--no-extras"]))
    assert rec["n_gpus"] == 1 and rec["steps"] == 5 and rec["unit"] == "atoms*steps/s" and rec["value"] > 0
    assert rec["config"]["graphs_per_gpu"] == 100
    # the dominant kernel of the default (split-f16) path, with the fp32-MFMA block launch beside it
    assert rec["roofline"]["kernel"].startswith("forward_mega_kernel") and 0 < rec["roofline"]["frac"] < 1
    assert rec["roofline"]["f32_mfma"]["kernel"].startswith("layer_combo_kernel")
    assert rec["f32_mfma_ms_per_step"] > rec["ms_per_step"] > 0
    assert rec["cpu_baseline"] is None and "c5" not in rec


def test_sample_sharded_under_torchrun_one_rank(tmp_path):
    """configs[2]'s harness: sharded sampling, rank 0 gathers and writes samples_all.pkl in the reference's result
    format; the file loads through tsdiff_amd.io and the run is reproducible for a seed"""
    from tsdiff_amd import io as tio
    outs = []
    for k in range(2):
        d = tmp_path / f"run{k}"
        so = _torchrun(["tools/sample_sharded.py", "--graphs", "8", "--models", "2", "--batch-size", "5", "--steps", "3",
                        "--seed", "41", "--out", str(d)])
        rec = _json_line(so)
        assert rec["graphs"] == 8 and rec["gpus"] == 1 and rec["checkpoints"] == 2 and rec["steps"] == 3 and rec["seed"] == 41
        res = tio.load_samples(os.path.join(str(d), "samples_all.pkl"))
        assert len(res) == 8
        for r in res:
            assert r.pos_gen.shape == r.pos.shape and torch.isfinite(r.pos_gen).all()
            assert r.edge_index.shape[0] == 2 and r.edge_type.shape[0] == r.edge_index.shape[1]
        outs.append(torch.cat([r.pos_gen for r in res]))
    assert torch.equal(outs[0], outs[1])  # same seed, same GPU count: the same samples
