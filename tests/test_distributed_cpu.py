"""CPU: the N > 1 path with world_size 2 over gloo (two real processes, 127.0.0.1
rendezvous): broadcast, allgather_object, the bucketed gradient reducer, data-parallel ==
large-batch equivalence, and the trainer's DP rules."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_world_size_two_gloo():
    port = str(_free_port())
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="2",
                   CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py")],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out}"
        assert f"rank {rank} OK" in out


def test_bench_bare_launch_spawns_ranks_and_relays_failure():
    """`python bench.py --gpus 2` without a launcher: the parent must spawn two ranks with a working rendezvous and, when
    the ranks fail (no GPU here: "bench.py needs an MI355X"), exit non-zero instead of hanging or printing a line."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    assert r.stderr.count("needs an MI355X") >= 1 and "launch with torch.distributed.run" not in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
