"""The N-rank harness of bench.py on CPU (gloo), kernels skipped (`--dry-run`): `python bench.py --gpus N` must start
N rank processes by itself, shard both scaling modes, time with barriers + max over ranks, all-reduce the gradient
bucket in the product's hook order (dp.OverlappedGradReducer: "rest" beside k_wgrad_emb, then "embedding", both
waited for before the next weight re-pack) and print ONE rank-0 JSON line (BASELINE.json configs[3])."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIVE = 7083151          # elements of the 19 live tensors at Yelp dims (V=9353, E=300, H=512, L=2)
EMB = 9353 * 300


def _bench(*args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=e, capture_output=True,
                          text=True, timeout=timeout)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout                       # ONE line, from rank 0 only
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_weak_scaling():
    r = _bench("--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr
    j = _json_line(r.stdout)
    assert j["dry_run"] is True and j["n_gpus"] == 2 and j["scaling"] == "weak"
    assert j["config"]["parallelism"] == "dp2" and j["config"]["global_batch"] == 2048 and j["local_batch"] == 1024
    assert j["steps"] == 3 and j["warmup"] == 1 and j["value"] > 0 and j["ms_per_step"] > 0
    assert j["bucket_allreduce_ok"] is True and j["bucket_numel"] == LIVE
    # the product's message order: everything but the embedding first, the embedding (first in the bucket) last
    assert j["messages_per_step"] == [["rest", LIVE - EMB], ["embedding", EMB]]
    assert "two messages" in j["unfrozen"]["what"] and j["dtype"] == "bf16"
    assert j["roofline"] is None and j["cpu_baseline"] is None      # nothing measured in a dry run


def test_strong_scaling_shards_one_global_batch_and_single_message_mode():
    r = _bench("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1", "--scaling", "strong", "--no-overlap")
    assert r.returncode == 0, r.stderr
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong"
    assert j["config"]["global_batch"] == 1024 and j["local_batch"] == 512
    assert j["bucket_allreduce_ok"] is True and "one message" in j["unfrozen"]["what"]


def test_yelp64_dtype_field_follows_the_soft_token_dtype():
    r = _bench("--dry-run", "--steps", "2", "--warmup", "1", "--workload", "yelp64")
    assert r.returncode == 0, r.stderr
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 1 and j["config"]["parallelism"] == "dp1" and j["config"]["global_batch"] == 64
    assert j["dtype"].startswith("bf16 (fp32 soft tokens")


def test_world_size_mismatch_is_refused():
    r = _bench("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"}, timeout=60)
    assert r.returncode != 0
    assert "--gpus 2 but WORLD_SIZE=4" in r.stderr


def test_under_a_launcher_the_ranks_come_from_the_environment():
    """What the driver does: torch.distributed.run sets RANK / WORLD_SIZE / MASTER_*; bench.py must not spawn again."""
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "dp2" and j["bucket_allreduce_ok"] is True
