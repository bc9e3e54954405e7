"""CPU, world_size 2 over gloo: the data-parallel contract of the step.

The gradient bucket logic is device-agnostic plumbing (flat buffer + one all-reduce + 1/world scaling);
here it is driven with gradients produced by the oracle on each rank's shard and checked against the
single-process gradient of the concatenated global batch (what nn.DataParallel computes in the reference,
lib/trainer.py:162-166,245-250).  BatchNorm-free sub-problem (S only + a linear head) so that per-rank
statistics do not enter; the BN-per-rank semantics are documented in DESIGN.md §5."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import wgs_oracle as O
from tests import golden_inputs as GI
from warpedganspace_amd.trainer import FlatBucket


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _loss(sd, head, z, idx, mag, gamma, K):
    mask = GI.one_hot(idx, K)
    shift = mag.reshape(-1, 1) * O.support_sets_forward(sd, mask, z, True, gamma)
    logits = (z + shift) @ head
    return O.training_loss(logits, shift.norm(dim=1), idx, mag)[0]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    K, N, d, Bg = 8, 2, 16, 8
    c = GI.support_sets_case(K, N, d, Bg, 77, learn_gammas=True)
    head = GI.rt(78, d, K)
    # this rank's replica of the trainable parameters, re-homed into a flat bucket
    table = torch.nn.Parameter(c['sd']['SUPPORT_SETS'].clone())
    lg = torch.nn.Parameter(c['sd']['LOGGAMMA'].clone())
    bucket = FlatBucket([(1e-4, [table]), (1e-4, [lg])], torch.device('cpu'))   # two groups, like [R | S] in TrainStep
    sl = slice(rank * Bg // world, (rank + 1) * Bg // world)          # shard the global batch by sample
    sd = {'SUPPORT_SETS': table, 'ALPHAS': c['sd']['ALPHAS'], 'LOGGAMMA': lg}
    loss = _loss(sd, head, c['z'][sl], c['idx'][sl], c['gout'][sl, 0] * 0.3, c['gamma'], K)
    g_table, g_lg = torch.autograd.grad(loss, [table, lg])
    bucket.zero_grad()
    bucket.gview[id(table)].copy_(g_table)
    bucket.gview[id(lg)].copy_(g_lg)
    # the step's collectives exactly as TrainStep.step issues them: one async all-reduce (sum) per bucket group, on
    # views of the flat gradient, waited for before Adam
    pending = [dist.all_reduce(bucket.grad[a:b], async_op=True) for _, a, b in bucket.groups]
    for w in pending:
        w.wait()
    avg = bucket.grad / world                                          # Adam's grad_scale = 1/world
    if rank == 0:
        q.put((avg.clone(), table.data_ptr() == bucket.flat.data_ptr()))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_allreduced_bucket_equals_global_batch_gradient():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    avg, aliased = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert aliased                                                      # parameters are views of the flat buffer
    K, N, d, Bg = 8, 2, 16, 8
    c = GI.support_sets_case(K, N, d, Bg, 77, learn_gammas=True)
    head = GI.rt(78, d, K)
    table = c['sd']['SUPPORT_SETS'].clone().requires_grad_(True)
    lg = c['sd']['LOGGAMMA'].clone().requires_grad_(True)
    sd = {'SUPPORT_SETS': table, 'ALPHAS': c['sd']['ALPHAS'], 'LOGGAMMA': lg}
    loss = _loss(sd, head, c['z'], c['idx'], c['gout'][:, 0] * 0.3, c['gamma'], K)   # mean over the GLOBAL batch
    g_table, g_lg = torch.autograd.grad(loss, [table, lg])
    ref = torch.cat([g_table.reshape(-1), g_lg.reshape(-1)])
    assert (avg[:ref.numel()] - ref).abs().max().item() < 1e-5 * ref.abs().max().item() + 1e-9


def test_flat_bucket_keeps_conv_memory_layout():
    conv = torch.nn.Conv2d(6, 4, 3, bias=False)
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    w0 = conv.weight.detach().clone()
    lin = torch.nn.Linear(5, 3)
    b = FlatBucket([(1e-3, [conv.weight, lin.weight, lin.bias])], torch.device('cpu'))
    assert torch.equal(conv.weight.detach(), w0)
    assert conv.weight.permute(0, 2, 3, 1).is_contiguous()               # still [Co,kh,kw,Ci] in memory (packed)
    assert b.gview[id(conv.weight)].shape == (4, 9, 6)
    b.gview[id(conv.weight)].fill_(1.0)
    assert float(conv.weight.grad.sum()) == conv.weight.numel()
    assert all(off % 4 == 0 for _, off, _ in b.segments)                 # 16-byte aligned segments


def test_host_pinning_plan_gives_disjoint_slices():
    """One launch thread per rank: ranks that share a NUMA node get disjoint, equal slices of its CPUs."""
    from warpedganspace_amd import hostpin as HP
    assert HP._cpulist('0-3,8,10-11\n') == {0, 1, 2, 3, 8, 10, 11}
    node, allowed = set(range(0, 48)), set(range(0, 96))
    got = [HP.plan(node, allowed, 4, s) for s in range(4)]
    assert all(len(g) == 12 for g in got) and len(set().union(*got)) == 48
    assert all(not (got[i] & got[j]) for i in range(4) for j in range(i))
    assert HP.plan({200, 201}, {0, 1, 2, 3}, 2, 1) == {2, 3}                      # node CPUs outside the allowed set: fall back to the allowed ones
    assert len(HP.plan({0}, {0}, 8, 5)) == 1                                      # more ranks than CPUs: still one CPU each
    r = HP.pin_rank(0, 1)                                                         # no GPU here: reports why, changes nothing
    assert r['pinned'] is False and r['why']
    # the affinity covers the whole process (RCCL proxy, autograd's device thread): slices below MIN_CPUS_PER_RANK are not applied
    cpus, why = HP.choose(set(range(0, 48)), set(range(0, 96)), 4, 1)
    assert cpus == set(range(12, 24)) and 'own slice' in why
    cpus, why = HP.choose(set(range(0, 8)), set(range(0, 16)), 4, 1)              # 2 CPUs each: the whole node instead, shared
    assert cpus == set(range(0, 8)) and 'confined to the NUMA node' in why
    cpus, why = HP.choose(set(range(0, 8)), set(range(0, 8)), 8, 3)               # 8 ranks on 8 CPUs, one node: leave it alone
    assert cpus is None and 'left alone' in why
    os.environ['WGS_NO_PIN'] = '1'
    try:
        assert HP._plan_rank(0, 2, set(range(64)), [(0, 'a'), (0, 'b')])['why'] == 'WGS_NO_PIN=1'
    finally:
        del os.environ['WGS_NO_PIN']
    assert len(HP.plan_all(8)) == 8


class _FakeWriter:
    def __init__(self):
        self.n = 0

    def add_scalar(self, *a):
        self.n += 1


def _tb_worker(rank, world, port, root, q):
    """The statistics branch of Trainer.train's loop as every rank takes it (with --tensorboard the writer exists on
    rank 0 only, and pop_stats() is a collective)."""
    import types
    from warpedganspace_amd.trainer import Trainer
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    p = types.SimpleNamespace(tensorboard=False, log_freq=3, max_iter=6, batch_size=4)
    tr = Trainer(params=p, exp_dir='exp', use_cuda=False, root=os.path.join(root, 'r%d' % rank))
    if rank == 0:
        tr.tb_writer = _FakeWriter()            # as `--tensorboard` leaves it: a writer on rank 0, none elsewhere
    before = tr.tb_writer is not None           # what the old loop condition looked at: differs between the ranks
    tr.agree_tensorboard()
    n_coll = 0
    for it in range(1, 2 * p.log_freq + 1):
        if tr.stats_due(it):                    # pop_stats(): one all-reduce per call
            t = torch.ones(1)
            dist.all_reduce(t)
            assert float(t) == world
            n_coll += 1
    dist.barrier()
    q.put((rank, before, tr.tb_active, n_coll))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_every_rank_pops_statistics_in_the_same_iterations(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_tb_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(90)
        assert p.exitcode == 0                  # a mismatched collective would hang until the join times out
    res = sorted(q.get() for _ in range(world))
    assert [r[1] for r in res] == [True, False]           # only rank 0 owns a writer ...
    assert [r[2] for r in res] == [True, True]            # ... every rank knows it after agree_tensorboard()
    assert [r[3] for r in res] == [6, 6]                  # statistics popped in every iteration, on BOTH ranks
