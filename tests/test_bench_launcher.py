"""CPU: `python bench.py --gpus N` starts N ranks itself (torch.distributed.run on 127.0.0.1) when no launcher did,
relays ONE JSON line from rank 0 with n_gpus == N, and reduces the ranks' count vectors with one all-reduce
(deepdish_amd/multistream.reduce_counts -- the same call the GPU run makes over RCCL).  --rehearse-cpu swaps the
backend for gloo and processes no frames: this checks the launcher and the reduction, not throughput."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR')}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, capture_output=True, text=True, env=env,
                          timeout=timeout, cwd=ROOT)


def test_gpus_2_spawns_two_ranks_and_reduces_counts():
    r = _run(['--gpus', '2', '--rehearse-cpu'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                     # exactly one JSON line on stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['value'] is None and 'rehearsal' in out
    assert out['counts_pos_neg_int_del'] == [1 + 2, 10 + 20, 11 + 22, 0]       # rank r contributes (r+1) * (1, 10, 11, 0)


def test_gpus_must_match_the_launcher_world_size():
    r = _run(['--gpus', '4', '--rehearse-cpu'], env_extra={'RANK': '0', 'LOCAL_RANK': '0', 'WORLD_SIZE': '2',
                                                          'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29999'})
    assert r.returncode == 2 and 'WORLD_SIZE=2' in r.stderr


def test_single_rank_rehearsal():
    r = _run(['--rehearse-cpu'])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip())
    assert out['n_gpus'] == 1 and out['counts_pos_neg_int_del'] == [1, 10, 11, 0]


def test_gpus_8_rehearsal_keeps_the_node_within_its_cores():
    """The driver's scaling run is N = 1, 2, 4, 8 on one node: eight ranks come up, one JSON line, n_gpus == 8, counts summed
    over all eight, and what the ranks would start on the host (generator processes, worker-group threads, host-pool
    threads) is each rank's share of the node's cores -- generators cores // (2 * 8), never fewer than one of anything."""
    r = _run(['--gpus', '8', '--rehearse-cpu'], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 8 and out['value'] is None
    tri = sum(range(1, 9))
    assert out['counts_pos_neg_int_del'] == [tri, 10 * tri, 11 * tri, 0]
    cores = out['host_cores']
    ranks = out['per_rank_host']
    assert sorted(b['rank'] for b in ranks) == list(range(8))
    for b in ranks:
        assert b['generator_processes'] == max(1, min(16, cores // 16))
        assert b['host_pool_threads'] == max(1, min(12, cores // 8 - b['worker_groups'] - 1))
        assert b['worker_groups'] >= 1


def test_ranks_started_by_an_external_launcher():
    """The driver starts the ranks itself (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`): bench.py must then NOT spawn again, must put HSA_ENABLE_IPC_MODE_LEGACY=0 into its own
    environment before torch is imported (RCCL's dmabuf IPC on this host driver), and rank 0 reports the communicator's size."""
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR', 'HSA_ENABLE_IPC_MODE_LEGACY')}
    env['DD_BENCH_REPORT_ENV'] = '1'
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--rehearse-cpu'],
                       capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['rccl_ranks'] == 2 and out['collective_backend'] == 'gloo'
    assert out['counts_pos_neg_int_del'] == [3, 30, 33, 0]
    assert out['env_of_rank0']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
