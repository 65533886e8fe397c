"""bench.py's N-rank launch path (the reference starts one process per GPU: Downstream/Text/script/adapter_houlsby.py:58-59,
run.py:685) exercised WITHOUT a GPU: A4R_BENCH_CONTROL_ONLY=1 keeps the self-launch, rendezvous on 127.0.0.1, rank accounting
and the single-JSON-line contract and skips the compute; the backend is gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, extra_env=None, drop=()):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT') + tuple(drop)}
    env.update(A4R_BENCH_CONTROL_ONLY='1', A4R_BENCH_BACKEND='gloo')
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True, timeout=600)


def json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith('{')]


def test_self_launch_two_ranks():
    r = run(['--gpus', '2', '--steps', '1', '--warmup', '0'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                      # ONE line, printed by rank 0
    assert lines[0]['n_gpus'] == 2 and lines[0]['rccl_ranks'] == 2


def test_single_rank_needs_no_launcher():
    r = run(['--gpus', '1'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]['n_gpus'] == 1 and lines[0]['rccl_ranks'] == 1


def test_rank_count_mismatch_is_an_error():
    """Under a launcher that gives fewer ranks than --gpus the run must fail instead of printing n_gpus of a job that did not run."""
    r = run(['--gpus', '4'], extra_env=dict(WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'))
    assert r.returncode != 0
    assert not json_lines(r.stdout)
