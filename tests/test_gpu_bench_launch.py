"""-m gpu: `python bench.py --gpus 2` on a 1-GPU box.  M2V_BENCH_SHARE_GPU=1 puts both ranks on GPU 0 and M2V_DIST_BACKEND=gloo
replaces RCCL (which refuses two ranks on one device) - the launch path, the per-rank work, the barrier / max-over-ranks
timing and, in strips mode, the halo exchange + gather + assembly and the whole-stream oracle check are the real ones."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"M2V_BENCH_SHARE_GPU": "1", "M2V_DIST_BACKEND": "gloo"})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return lines[0]


def test_sequences_mode_two_ranks():
    line = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--prewarm", "0.2", "--no-cpu-baseline", "--gops", "2", "--strips-steps", "6", "--long-gops", "6"])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["scaling"] == "weak"
    assert line["config"]["launched_by"] == "bench.py" and line["config"]["dist_backend"] == "gloo"
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
    assert line["parity_check"]["ranks_checked"] == 2 and line["parity_check"]["identical_to_oracle"] is True     # every rank, its own clip
    # the ONE line also carries config c5: both strips legs ran as fresh ranks after the c4 ranks had exited.  On this hook (gloo, one GPU)
    # the RCCL leg runs the Python loop over gloo; the peer leg runs the NATIVE loop - landing blocks shared between the two processes through
    # hipIpc handles, sizes and strips through a communicator over gloo - two sequences in flight per rank, the output rank rotating
    st = line["strips"]
    for name in ("rccl", "peer"):
        assert "error" not in st[name], st[name]
        assert st[name]["value"] > 0 and st[name]["n_gpus"] == 2 and st[name]["identical_to_oracle"] is True, st[name]
    assert st["rccl"]["strip_loop"] == "python"
    assert st["peer"]["strip_loop"] == "native" and st["peer"]["transport"].startswith("peer+") and st["peer"]["sequences_in_flight"] == 3
    assert st["peer"]["in_flight_output_rank_rotating"]["value"] > 0 and st["peer"]["peer_sequences"] + st["peer"]["giveups"] > 0
    long = st["peer"]["in_flight_long_sequence"]
    assert long["gops"] == 6 and long["frames"] == 54 and long["value"] > 0 and long["gop_steps_ran_as"] in ("peer", "calls")
    print("strips legs on the shared-GPU hook:", {k: (v["value"], v["ms_per_sequence"], v.get("giveups")) for k, v in st.items()})


def test_the_drivers_form_torch_distributed_run_also_measures_the_strips():
    """python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2: the ranks are somebody else's children and have touched the GPU
    when their c4 work is done - each then starts ONE child per strips leg (never an exec) and rank 0 prints the c4 line with `strips`, last"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"M2V_BENCH_SHARE_GPU": "1", "M2V_DIST_BACKEND": "gloo"})
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29641", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--prewarm", "0.2",
                        "--gops", "2", "--strips-steps", "4", "--long-gops", "4", "--sustain", "0"], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = lines[0]
    assert line["n_gpus"] == 2 and line["config"]["launched_by"] == "caller" and line["parity_check"]["identical_to_oracle"] is True
    st = line["strips"]
    assert "error" not in st["rccl"] and "error" not in st["peer"], st
    assert st["rccl"]["identical_to_oracle"] is True and st["peer"]["identical_to_oracle"] is True
    assert st["peer"]["strip_loop"] == "native" and st["peer"]["n_gpus"] == 2


def test_strips_mode_two_ranks_whole_stream_parity():
    line = run_bench(["--gpus", "2", "--mode", "strips", "--steps", "2", "--warmup", "1", "--prewarm", "0", "--gops", "2"])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["scaling"] == "strong"
    assert line["config"]["strip_loop"] == "python"            # gloo between two processes on one GPU: the reference loop
    assert line["parity_check"]["identical_to_oracle"] is True and line["parity_check"]["gops_compared"] == 2


def test_strips_mode_two_ranks_peer_transport_between_processes_in_flight():
    line = run_bench(["--gpus", "2", "--mode", "strips", "--transport", "peer", "--rotate-dst", "--steps", "6", "--warmup", "2", "--prewarm", "0", "--gops", "2"])
    assert line["n_gpus"] == 2 and line["config"]["strip_loop"] == "native" and line["config"]["transport"] == "peer+callbacks"
    assert line["sequences_in_flight"] == 3 and line["in_flight_output_rank_0"]["identical_to_the_blocking_call"] is True
    assert line["in_flight_output_rank_rotating"]["value"] > 0
    assert line["parity_check"]["identical_to_oracle"] is True
    assert len(line["per_rank_ms_per_step"]["ranks"]) == 2


def test_strips_mode_one_rank_runs_the_native_loop():
    line = run_bench(["--gpus", "1", "--mode", "strips", "--steps", "2", "--warmup", "1", "--prewarm", "0", "--gops", "2"])
    assert line["n_gpus"] == 1 and line["config"]["strip_loop"] == "native"
    assert line["parity_check"]["identical_to_oracle"] is True


def test_config_c2_line():
    line = run_bench(["--config", "c2", "--steps", "3", "--warmup", "1", "--prewarm", "0.2", "--gops", "64", "--no-e2e"])
    assert line["config"]["frames"] == 64 and "I only" in line["metric"]
    assert line["parity_check"]["identical_to_oracle"] is True
    assert "k_mb<1,false>" in line["roofline"]["kernel"] and line["roofline"]["frac"] > 0
