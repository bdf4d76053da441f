"""Multi-rank tests, world_size 2 over `gloo`.

CPU (not gpu): the partition/layout/exchange plumbing of seoul_tourism_recommendation_ngcf_amd/dist.py
(ShardLayout, slab_coo, allgather_rows, owner_rows_sum) drives a two-rank propagation whose per-slab
arithmetic is done by the ORACLE inside the test; the result must equal the unsharded oracle.  This checks
that the exchange is correct by construction without any CPU compute path in the product.

GPU (-m gpu): two ranks share cuda:0 (RCCL refuses two ranks on one device, so the process group is gloo
on device tensors); `ShardedPropagation` runs the real HIP kernels in both exchange schemes and must match
the single-GPU engine.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ngcf_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy_graph(n_user, n_item, n_inter, seed):
    """Same generator as graphs.synthetic_bipartite, on the CPU generator."""
    from seoul_tourism_recommendation_ngcf_amd import graphs
    return graphs.synthetic_bipartite(n_user, n_item, n_inter, seed=seed, device="cpu")


def _params(d0, layers, seed):
    g = torch.Generator().manual_seed(seed)
    dims = [d0] + list(layers)
    w1 = [(torch.rand((dims[k + 1], dims[k]), generator=g) - 0.5) * 0.3 for k in range(len(layers))]
    w2 = [(torch.rand((dims[k + 1], dims[k]), generator=g) - 0.5) * 0.3 for k in range(len(layers))]
    b1 = [(torch.rand((dims[k + 1],), generator=g) - 0.5) * 0.1 for k in range(len(layers))]
    b2 = [(torch.rand((dims[k + 1],), generator=g) - 0.5) * 0.1 for k in range(len(layers))]
    return w1, b1, w2, b2


def _dense_oracle(LE, E, W1, b1, W2, b2):
    M = torch.nn.functional.linear(LE, W1, b1) + torch.nn.functional.linear(E, W1, b1) \
        + torch.nn.functional.linear(LE * E, W2, b2)
    c = torch.nn.functional.leaky_relu(M, 0.2)
    return c, torch.nn.functional.normalize(c, p=2, dim=1)


def _cpu_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from seoul_tourism_recommendation_ngcf_amd import dist as nd
        U, I, d0, layers = 300, 40, 16, (16, 12)
        coo = _toy_graph(U, I, 3000, seed=5)
        rows, cols, vals = coo["rows"], coo["cols"], coo["vals"]
        N = U + I
        g = torch.Generator().manual_seed(1)
        E0 = torch.randn((N, d0), generator=g)
        w1, b1, w2, b2 = _params(d0, layers, 2)
        L = torch.sparse_coo_tensor(torch.stack([rows, cols]), vals, (N, N))
        want = orc.propagate_torch(L, E0[:U], E0[U:], w1, b1, w2, b2)

        # ---- all-gather scheme: padded chunk-major / rank-major layout, the user slab pipelined in 3 row chunks, every
        # chunk's asynchronous all-gather issued right after the chunk (dist._propagate_allgather does exactly this);
        # the slabs are cut from the interaction triplets, never from the doubled COO
        from seoul_tourism_recommendation_ngcf_amd import graphs
        iu, ii, iw = graphs.synthetic_interactions(U, I, 3000, seed=5, device="cpu")
        iv, deg_u, deg_i = nd.laplacian_values(iu, ii, iw, U, I)
        cnt = torch.cat([deg_u, deg_i])
        assert torch.equal(cnt, nd.row_counts(rows, N))
        ub = nd.balanced_bounds(cnt, 0, U, world)
        ib = nd.balanced_bounds(cnt, U, N, world)
        C = 3
        lay = nd.ShardLayout(U, I, ub, ib, nd.chunk_bounds(cnt, ub, C))
        assert lay.chunks == C and lay.P == world * (C * lay.mc + lay.mi)
        pos_all = lay.to_padded(torch.arange(N))
        assert pos_all.unique().numel() == N and int(pos_all.max()) < lay.P
        (sur, suc, suv), (sir, sic, siv) = nd.cut_slabs(iu, ii, iv, U, ub[rank], ub[rank + 1], ib[rank] - U, ib[rank + 1] - U)
        for (a_, b_, c_), (lo_, hi_) in (((sur, suc, suv), (ub[rank], ub[rank + 1])), ((sir, sic, siv), (ib[rank], ib[rank + 1]))):
            r_, c2_, v_ = nd.slab_coo(rows, cols, vals, lo_, hi_)           # the same rows of the full COO, entry for entry
            assert torch.equal(a_ - lo_, r_) and torch.equal(b_, c2_) and torch.equal(c_, v_)
        full = torch.full((lay.P, d0), float("nan"))
        full[pos_all] = E0
        nu, ni = lay.n_users_of(rank), lay.n_items_of(rank)
        blocks_u, blocks_i = [E0[ub[rank]:ub[rank + 1]]], [E0[ib[rank]:ib[rank + 1]]]

        def slab_layer(r_, c_, v_, lo, hi, pos, m, k):
            n_own = hi - lo
            Ls = torch.sparse_coo_tensor(torch.stack([r_ - lo, lay.to_padded(c_)]), v_, (n_own, lay.P))
            table = torch.nan_to_num(full, nan=0.0)               # padding rows are never referenced
            carry, nrm = _dense_oracle(torch.mm(Ls, table), full[pos:pos + n_own], w1[k], b1[k], w2[k], b2[k])
            send = torch.full((m, carry.shape[1]), float("nan"))
            send[:n_own] = carry
            return send, nrm

        for k in range(len(layers)):
            nxt = torch.full((lay.P, layers[k]), float("nan"))
            works, keep = [], []
            send, nrm_i = slab_layer(sir, sic, siv, ib[rank], ib[rank + 1], lay.item_pos(rank), lay.mi, k)
            works.append(nd.allgather_rows(nxt[lay.n_user_pos:], send, async_op=True))
            keep.append(send)
            nrm_u = []
            for j in range(C):
                lo, hi = lay.chunk_range(rank, j)
                a, b = (int(x) for x in torch.searchsorted(sur, torch.tensor([lo, hi])))
                send, nrm = slab_layer(sur[a:b], suc[a:b], suv[a:b], lo, hi, lay.user_pos(rank, j), lay.mc, k)
                ra, rb = lay.chunk_region(j)
                works.append(nd.allgather_rows(nxt[ra:rb], send, async_op=True))
                keep.append(send)
                nrm_u.append(nrm)
            for wk in works:
                wk.wait()
            blocks_u.append(torch.cat(nrm_u, 0))
            blocks_i.append(nrm_i)
            full = nxt
            # every real node's row arrived; only padding rows stay NaN
            assert not torch.isnan(full[pos_all]).any()
        got_u, got_i = torch.cat(blocks_u, 1), torch.cat(blocks_i, 1)
        assert torch.allclose(got_u, want[ub[rank]:ub[rank + 1]], atol=1e-6)
        assert torch.allclose(got_i, want[ib[rank]:ib[rank + 1]], atol=1e-6)
        # owner-served row gathers: every rank ends with all rows
        u_id = torch.tensor([0, U - 1, 17, ub[1], ub[1] - 1, 5])
        ow, loc = lay.owner_of_user(u_id)
        mine = ow == rank
        local = got_u[torch.where(mine, loc, torch.zeros_like(loc))]
        assert torch.allclose(nd.owner_rows_sum(local, mine), want[:U][u_id], atol=1e-6)
        it = torch.tensor([0, I - 1, 3, ib[1] - U])
        ow, loc = lay.owner_of_item(it)
        mine = ow == rank
        local = got_i[torch.where(mine, loc, torch.zeros_like(loc))]
        assert torch.allclose(nd.owner_rows_sum(local, mine), want[U:][it], atol=1e-6)

        # ---- bipartite scheme (dist._propagate_bipartite does exactly this): users partitioned by stored entries, item carry
        # replicated in the owner-major padded numbering; per layer a reduce-scatter of the item partial sums to the owners
        # (gloo has none: all-reduce + own rows), the dense half for the owned items only, an all-gather of their carry rows
        eb = nd.balanced_bounds(cnt, 0, U, world)
        ob = nd.even_bounds(0, I, world)
        mi = max(ob[q + 1] - ob[q] for q in range(world))
        PI = world * mi
        ipos = nd.padded_item_pos(torch.arange(I), ob, mi)
        assert ipos.unique().numel() == I and int(ipos.max()) < PI
        lo, hi = eb[rank], eb[rank + 1]
        (sur, suc, suv), _ = nd.cut_slabs(iu, ii, iv, U, lo, hi, 0, 0)
        pos = nd.padded_item_pos(suc - U, ob, mi)
        Lu = torch.sparse_coo_tensor(torch.stack([sur - lo, pos]), suv, (hi - lo, PI))
        Lit = torch.sparse_coo_tensor(torch.stack([pos, sur - lo]), suv, (PI, hi - lo))
        eu = E0[lo:hi]
        ei = torch.full((PI, d0), float("nan"))
        ei[ipos] = E0[U:]
        own = slice(rank * mi, rank * mi + (ob[rank + 1] - ob[rank]))
        bu, bi = [eu], [E0[U + ob[rank]:U + ob[rank + 1]]]
        for k in range(len(layers)):
            part = torch.mm(Lit, eu)                                # partial sums of ALL items over the local users
            work = dist.all_reduce(part, async_op=True)
            cu, nu_ = _dense_oracle(torch.mm(Lu, torch.nan_to_num(ei, nan=0.0)), eu, w1[k], b1[k], w2[k], b2[k])
            work.wait()
            ci, ni_ = _dense_oracle(part[own], ei[own], w1[k], b1[k], w2[k], b2[k])      # the owned items only
            send = torch.full((mi, layers[k]), float("nan"))
            send[:ci.shape[0]] = ci
            nxt = torch.full((PI, layers[k]), float("nan"))
            nd.allgather_rows(nxt, send)
            assert not torch.isnan(nxt[ipos]).any()
            bu.append(nu_)
            bi.append(ni_)
            eu, ei = cu, nxt
        assert torch.allclose(torch.cat(bu, 1), want[lo:hi], atol=1e-6)
        assert torch.allclose(torch.cat(bi, 1), want[U + ob[rank]:U + ob[rank + 1]], atol=1e-6)
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def _cpu_train_worker(rank, world, port, ret):
    """Backward through the exchange on CPU tensors: the differentiable exchange steps of dist.py (AllGatherRows, ReduceScatterRows,
    OwnerRowsSum, _SumGrads) drive the bipartite scheme with the ORACLE's arithmetic (torch CPU ops + torch autograd) per slab; every
    parameter's gradient on every rank must equal the unsharded oracle's."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from seoul_tourism_recommendation_ngcf_amd import dist as nd, graphs
        U, I, d0, layers = 240, 31, 12, (12, 8)
        iu, ii, iw_ = graphs.synthetic_interactions(U, I, 2500, seed=6, device="cpu")
        iv, deg_u, deg_i = nd.laplacian_values(iu, ii, iw_, U, I)
        coo = _toy_graph(U, I, 2500, seed=6)
        N = U + I
        L = torch.sparse_coo_tensor(torch.stack([coo["rows"], coo["cols"]]), coo["vals"], (N, N))
        g = torch.Generator().manual_seed(3)
        E0 = torch.randn((N, d0), generator=g)
        w1, b1, w2, b2 = _params(d0, layers, 4)
        u_id, pos, neg = (torch.randint(0, hi, (40,), generator=g) for hi in (U, I, I))

        def leaves():
            ps = [E0[:U].clone(), E0[U:].clone()] + [t.clone() for t in w1 + b1 + w2 + b2]
            return [p.requires_grad_(True) for p in ps]

        # unsharded oracle
        ref = leaves()
        n = len(layers)
        all_E = orc.propagate_torch(L, ref[0], ref[1], ref[2:2 + n], ref[2 + n:2 + 2 * n], ref[2 + 2 * n:2 + 3 * n], ref[2 + 3 * n:])
        ou, op, on = orc.gather_torch(all_E, U, u_id, pos, neg)
        orc.bpr_torch(ou, op, on, 0.025, 40).backward()
        # sharded, same arithmetic per slab
        cnt = torch.cat([deg_u, deg_i])
        eb = nd.balanced_bounds(cnt, 0, U, world)
        ob = nd.even_bounds(0, I, world)
        mi = max(ob[q + 1] - ob[q] for q in range(world))
        PI = world * mi
        lo, hi = eb[rank], eb[rank + 1]
        ni = ob[rank + 1] - ob[rank]
        (sur, suc, suv), _ = nd.cut_slabs(iu, ii, iv, U, lo, hi, 0, 0)
        ppos = nd.padded_item_pos(suc - U, ob, mi)
        Lu = torch.sparse_coo_tensor(torch.stack([sur - lo, ppos]), suv, (hi - lo, PI))
        Lit = torch.sparse_coo_tensor(torch.stack([ppos, sur - lo]), suv, (PI, hi - lo))
        mine = leaves()
        outs = nd._SumGrads.apply(None, {0: eb}, *mine)       # the user table's gradient travels as an all-gather of the ranks' slabs
        uw, iw = outs[0], outs[1]
        pw1, pb1, pw2, pb2 = outs[2:2 + n], outs[2 + n:2 + 2 * n], outs[2 + 2 * n:2 + 3 * n], outs[2 + 3 * n:]
        src = torch.zeros(PI, dtype=torch.int64)
        src[nd.padded_item_pos(torch.arange(I), ob, mi)] = torch.arange(I)
        eu, ei = uw[lo:hi], iw[src]
        own = slice(rank * mi, rank * mi + ni)
        bu, bi = [eu], [ei[own]]
        for k in range(n):
            le_own = nd.ReduceScatterRows.apply(torch.mm(Lit, eu), None)
            cu, nu_ = _dense_oracle(torch.mm(Lu, ei), eu, pw1[k], pb1[k], pw2[k], pb2[k])
            ci, ni_ = _dense_oracle(le_own[:ni], ei[own], pw1[k], pb1[k], pw2[k], pb2[k])
            bu.append(nu_)
            bi.append(ni_)
            if k + 1 < n:
                ei = nd.AllGatherRows.apply(torch.cat([ci, torch.zeros((mi - ni, ci.shape[1]))], 0), None)
            eu = cu
        allE_u, allE_i = torch.cat(bu, 1), torch.cat(bi, 1)
        assert torch.allclose(allE_u, all_E[lo:hi].detach(), atol=1e-6) and torch.allclose(allE_i, all_E[U + ob[rank]:U + ob[rank + 1]].detach(), atol=1e-6)

        def served(table, bounds, idx):
            b = torch.tensor(bounds)
            ow = (torch.searchsorted(b, idx, right=True) - 1).clamp(0, world - 1)
            own_ = ow == rank
            loc = torch.where(own_, idx - b[ow], torch.zeros_like(idx))
            rows = table[loc] if table.shape[0] else torch.zeros((idx.numel(), table.shape[1]))
            return nd.OwnerRowsSum.apply(rows, own_, None)
        su, sp, sn = served(allE_u, eb, u_id), served(allE_i, ob, pos), served(allE_i, ob, neg)
        orc.bpr_torch(su, sp, sn, 0.025, 40).backward()
        for a, b_ in zip(mine, ref):
            assert a.grad is not None and torch.allclose(a.grad, b_.grad, atol=1e-6 + 1e-5 * float(b_.grad.abs().max())), (a.shape,)
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_backward_through_the_exchange_on_gloo_cpu(world):
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_cpu_train_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert dict(ret) == {r: 1 for r in range(world)}


@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_exchange_plumbing_on_gloo_cpu(world):
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_cpu_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert dict(ret) == {r: 1 for r in range(world)}


def test_layout_and_bounds_single_process():
    from seoul_tourism_recommendation_ngcf_amd import dist as nd
    cnt = torch.tensor([1, 1, 1, 1, 50, 2, 2, 2, 2, 40])          # 5 users, 5 items
    ub = nd.balanced_bounds(cnt, 0, 5, 2)
    ib = nd.balanced_bounds(cnt, 5, 10, 2)
    assert ub[0] == 0 and ub[-1] == 5 and ib[0] == 5 and ib[-1] == 10
    lay = nd.ShardLayout(5, 5, ub, ib)
    pos = lay.to_padded(torch.arange(10))
    assert len(set(pos.tolist())) == 10 and int(pos.max()) < lay.P
    for q in range(2):                                            # chunks are contiguous and rank-major
        assert pos[ub[q]:ub[q + 1]].tolist() == list(range(lay.user_pos(q), lay.user_pos(q) + lay.n_users_of(q)))
        assert pos[ib[q]:ib[q + 1]].tolist() == list(range(lay.item_pos(q), lay.item_pos(q) + lay.n_items_of(q)))
    # chunked layout: chunk-major, then rank-major; an empty chunk (repeated bound) holds nobody
    cl = nd.ShardLayout(5, 5, [0, 4, 5], [5, 9, 10], [0, 1, 1, 4, 4, 5, 5])          # W = 2, C = 3
    assert (cl.chunks, cl.mc, cl.mi, cl.P) == (3, 3, 4, 2 * (3 * 3 + 4))
    assert cl.to_padded(torch.arange(5)).tolist() == [0, 12, 13, 14, 9]          # user 4 = chunk 1 of rank 1
    assert cl.to_padded(torch.arange(5, 10)).tolist() == [18, 19, 20, 21, 22]
    assert cl.chunk_region(1) == (6, 12) and cl.user_pos(1, 2) == 15 and cl.n_users_of(0, 1) == 0
    assert nd.chunk_bounds(cnt, [0, 4, 5], 2)[::2] == [0, 4, 5]
    assert nd.even_bounds(0, 10, 4) == [0, 2, 5, 7, 10]
    r, c, v = nd.slab_coo(torch.tensor([0, 0, 2, 3, 3]), torch.arange(5), torch.ones(5), 2, 4)
    assert r.tolist() == [0, 1, 1] and c.tolist() == [2, 3, 4]


# ------------------------------------------------------------------------------------------------
# GPU: the real sharded HIP path, two ranks on one device
# ------------------------------------------------------------------------------------------------
def _gpu_worker(rank, world, port, mode, ret, backend="gloo", cabi=False, collectives="torch"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NGCF_P2P_TIMEOUT_MS="20000")
    # how the rows travel: "p2p" = the CU-free exchange (IPC-mapped buffers, copy streams, host-side waiting), "torch" =
    # torch.distributed collectives, "cabi" = ngcf_allgather_rows on the process group's communicator
    os.environ["NGCF_DIST_COLLECTIVES"] = "cabi" if cabi else collectives
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import seoul_tourism_recommendation_ngcf_amd as pkg
        from seoul_tourism_recommendation_ngcf_amd import dist as nd
        dev = torch.device("cuda:0")
        U, I = 6000, 500
        coo = pkg.graphs.synthetic_bipartite(U, I, 90000, seed=3, device=dev)
        num_dict = {"user": U, "item": I, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
        torch.manual_seed(7)
        model = pkg.NGCF(128, [128, 64], None, None, 1.0, [pkg.graphs.to_sparse_coo(coo)], num_dict, 64, dev).to(dev).eval()
        want = model.propagate(0).detach().clone()
        if mode == "allgather":
            # slabs cut from HOST interaction triplets (only this rank's part reaches the device), user slab in 3 chunks
            iu, ii, iw = (t.cpu() for t in pkg.graphs.synthetic_interactions(U, I, 90000, seed=3, device=dev))
            sh = nd.ShardedPropagation.from_interactions(model, iu, ii, iw, mode=mode, chunks=3, device=dev)
            assert sh.chunks == 3 and len(sh.csr_u) == 3 and (sh._cabi is not None) == cabi
            assert sum(c.nnz for c in sh.csr_u) + sh.csr_i.nnz <= coo["nnz"] // world + coo["nnz"] // 10
        else:
            sh = nd.ShardedPropagation.from_coo(model, coo["rows"], coo["cols"], coo["vals"], mode=mode)
        assert sh.backend == ("cabi" if cabi else collectives if world > 1 else "torch"), (sh.backend, getattr(sh, "p2p_error", None))
        with torch.no_grad():                                      # the inference path (with autograd recording, the bipartite scheme
            first = sh.propagate()                                 # would take its differentiable path: tested further down)
            first = (first[0].clone(), first[1].clone())
            au, ai = sh.propagate()                                # a second pass re-uses every buffer and both exchange regions
            assert torch.equal(au, first[0]) and torch.equal(ai, first[1])
            import time
            for it in range(4):                                    # ranks drifting apart: sleeps of 0 / 15 / 30 ms that rotate over the ranks,
                time.sleep(0.015 * ((rank + it) % 3))              # with and without a device sync - the sequence words and the
                if (rank + it) % 2:                                # acknowledgements keep producers and consumers in step
                    torch.cuda.synchronize()
                au, ai = sh.propagate()
                assert torch.equal(au, first[0]) and torch.equal(ai, first[1]), f"pass {it + 3} differs"
        assert torch.equal(au, first[0]) and torch.equal(ai, first[1])
        if mode == "bipartite":
            lo, hi = sh.ub[rank], sh.ub[rank + 1]
            ok = torch.allclose(au, want[lo:hi], atol=2e-5, rtol=2e-3) and \
                torch.allclose(ai, want[U + sh.ib[rank]:U + sh.ib[rank + 1]], atol=2e-5, rtol=2e-3)
        else:
            lay = sh.layout
            ok = torch.allclose(au, want[lay.ub[rank]:lay.ub[rank + 1]], atol=2e-5, rtol=2e-3) and \
                torch.allclose(ai, want[lay.ib[rank]:lay.ib[rank + 1]], atol=2e-5, rtol=2e-3)
        g = torch.Generator().manual_seed(11)
        u_id = torch.randint(0, U, (64,), generator=g).to(dev)
        pos = torch.randint(0, I, (64,), generator=g).to(dev)
        neg = torch.randint(0, I, (64,), generator=g).to(dev)
        u, p, n = sh.gather(u_id, pos, neg)
        ok = ok and torch.allclose(u, want[:U][u_id], atol=2e-5, rtol=2e-3) and torch.allclose(p, want[U:][pos], atol=2e-5, rtol=2e-3) \
            and torch.allclose(n, want[U:][neg], atol=2e-5, rtol=2e-3)
        loss = pkg.BPR(0.025, 64)(u, p, n)
        ref = orc.bpr_torch(want[:U][u_id].cpu(), want[U:][pos].cpu(), want[U:][neg].cpu(), 0.025, 64)
        ok = ok and abs(float(loss) - float(ref)) < 1e-4 * abs(float(ref))
        # the gathers are bit-exact copies of the OWNERS' rows (r04: over p2p every rank pulls every owner's block and picks): the
        # rows this rank owns can be checked against its own slab, and every later gather (the two exchange regions in turn) agrees
        if mode == "bipartite":
            lo_u, hi_u, lo_i, hi_i = sh.ub[rank], sh.ub[rank + 1], sh.ib[rank], sh.ib[rank + 1]
        else:
            lo_u, hi_u, lo_i, hi_i = sh.layout.ub[rank], sh.layout.ub[rank + 1], sh.layout.ib[rank] - U, sh.layout.ib[rank + 1] - U
        mu, mp_, mn = (u_id >= lo_u) & (u_id < hi_u), (pos >= lo_i) & (pos < hi_i), (neg >= lo_i) & (neg < hi_i)
        ok = ok and torch.equal(u[mu], au[u_id[mu] - lo_u]) and torch.equal(p[mp_], ai[pos[mp_] - lo_i]) and torch.equal(n[mn], ai[neg[mn] - lo_i])
        for it in range(3):
            time.sleep(0.01 * ((rank + it) % 2))
            u2, p2, n2 = sh.gather(u_id, pos, neg)
            ok = ok and torch.equal(u2, u) and torch.equal(p2, p) and torch.equal(n2, n)
        u3, p3, _ = sh.gather(u_id[:7], pos[:9], torch.empty(0))   # another shape, no negatives (experiment.py:82-91)
        ok = ok and torch.equal(u3, u[:7]) and torch.equal(p3, p[:9])
        big = [torch.cat([t, t.flip(0), t]) for t in (u_id, pos, neg)]      # a LARGER batch than any before: the gather exchange is rebuilt
        ub_, pb_, nb_ = sh.gather(*big)
        ok = ok and torch.equal(ub_[:64], u) and torch.equal(ub_[64:128], u.flip(0)) and torch.equal(pb_[128:], p) and torch.equal(nb_[:64], n)
        u4, p4, n4 = sh.gather(u_id, pos, neg)
        ok = ok and torch.equal(u4, u) and torch.equal(p4, p) and torch.equal(n4, n)
        if mode == "bipartite":
            # r04: nobody holds the previous result -> the same buffers are written again and E0 is not copied again; an in-place
            # update of a table (an optimizer step) is seen through its version counter
            del au, ai, u2, p2, n2, u3, p3, ub_, pb_, nb_, u4, p4, n4
            with torch.no_grad():
                sh.propagate()
                ptr = sh.allE_u.data_ptr()
                sh.propagate()
                ok = ok and sh.allE_u.data_ptr() == ptr and torch.equal(sh.allE_u, first[0]) and torch.equal(sh.allE_i, first[1])
                model.user_embedding.weight.add_(0.25)
                model.item_embedding.weight.mul_(0.5)
                a1 = sh.propagate()
                a1 = (a1[0].clone(), a1[1].clone())
                sh.invalidate_e0()
                a2 = sh.propagate()
                ok = ok and torch.equal(a1[0], a2[0]) and torch.equal(a1[1], a2[1]) and not torch.equal(a1[0], first[0])
                ok = ok and torch.equal(a2[0][:, :128], model.user_embedding.weight[lo_u:hi_u])
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("mode,world,collectives", [("bipartite", 2, "p2p"), ("allgather", 2, "p2p"), ("bipartite", 3, "p2p"),
                                                    ("allgather", 3, "p2p"), ("bipartite", 2, "torch"), ("allgather", 2, "torch"),
                                                    ("bipartite", 3, "torch"), ("allgather", 3, "torch")])
def test_sharded_propagation_ranks_share_one_gpu(mode, world, collectives):
    """Both exchange schemes over both transports, 2 and 3 ranks sharing the one GPU (process group gloo): "p2p" exchanges the rows
    through IPC-mapped exchange buffers pulled with device-to-device copies (include/ngcf_hip.h, ngcf_p2p_*), "torch" through
    torch.distributed; every rank's rows equal the single-GPU engine's, two passes give the same bits."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_gpu_worker, args=(world, _free_port(), mode, ret, "gloo", False, collectives), nprocs=world, join=True)
        assert dict(ret) == {r: True for r in range(world)}


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["bipartite", "allgather"])
def test_sharded_propagation_one_rank_over_rccl(mode):
    """The collectives of the sharded path issued through RCCL itself (backend "nccl"), with the one rank a one-GPU
    box allows: communicator set-up, async work handles on device buffers, all_gather_into_tensor on views."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_gpu_worker, args=(1, _free_port(), mode, ret, "nccl"), nprocs=1, join=True)
        assert dict(ret) == {0: True}


@pytest.mark.gpu
def test_allgather_rows_c_entry_point_on_the_process_groups_communicator():
    """ngcf_allgather_rows (include/ngcf_hip.h) driving ncclAllGather itself on the RCCL communicator of the process group,
    on a stream of its own: the chunked all-gather scheme end to end with the one rank a one-GPU box allows."""
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_gpu_worker, args=(1, _free_port(), "allgather", ret, "nccl", True), nprocs=1, join=True)
        assert dict(ret) == {0: True}


def _gpu_train_worker(rank, world, port, ret, collectives="torch"):
    """`loss.backward()` through the sharded propagation (experiment.py:57 across ranks): every parameter's gradient on every rank
    equals the single-GPU engine's gradient on the same graph, parameters and batch."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NGCF_DIST_COLLECTIVES=collectives, NGCF_P2P_TIMEOUT_MS="20000")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import seoul_tourism_recommendation_ngcf_amd as pkg
        from seoul_tourism_recommendation_ngcf_amd import dist as nd
        dev = torch.device("cuda:0")
        U, I, B = 5000, 330, 96
        coo = pkg.graphs.synthetic_bipartite(U, I, 70000, seed=4, device=dev)
        num_dict = {"user": U, "item": I, "sex": 2, "age": 76, "month": 13, "day": 32, "dayofweek": 7}
        g = torch.Generator().manual_seed(12)
        u_id, pos, neg = (torch.randint(0, hi, (B,), generator=g).to(dev) for hi in (U, I, I))
        u_id[:10] = u_id[10:20]                                    # duplicates in the batch
        grads = []
        for sharded in (False, True):
            torch.manual_seed(7)
            model = pkg.NGCF(65, [65, 64], None, None, 1.0, [pkg.graphs.to_sparse_coo(coo)], num_dict, B, dev).to(dev).eval()
            crit = pkg.BPR(0.025, B)
            if sharded:
                sh = nd.ShardedPropagation.from_coo(model, coo["rows"], coo["cols"], coo["vals"], mode="bipartite")
                assert sh.backend == collectives, (sh.backend, getattr(sh, "p2p_error", None))
                sh.propagate()
                assert isinstance(sh._train_comm(), nd.P2PCollectives) == (collectives == "p2p")
                u, p, n = sh.gather(u_id, pos, neg)
            else:
                all_E = model.propagate(0)
                u, p, n = all_E[:U][u_id], all_E[U:][pos], all_E[U:][neg]
            loss = crit(u, p, n)
            loss.backward()
            grads.append((float(loss), {k: v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None}))
        assert abs(grads[0][0] - grads[1][0]) <= 1e-5 * abs(grads[0][0])
        assert set(grads[0][1]) == set(grads[1][1]) and "user_embedding.weight" in grads[0][1] and "w2_list.1.bias" in grads[0][1]
        ok = True
        for k, want in grads[0][1].items():
            got = grads[1][1][k]
            scale = float(want.abs().max()) + 1e-12
            ok = ok and bool(((got - want).abs() <= 2e-3 * scale * 1e-1 + 2e-3 * want.abs()).all())
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,collectives", [(2, "torch"), (3, "torch"), (2, "p2p"), (3, "p2p")])
def test_backward_through_the_sharded_propagation_ranks_share_one_gpu(world, collectives):
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_gpu_train_worker, args=(world, _free_port(), ret, collectives), nprocs=world, join=True)
        assert dict(ret) == {r: True for r in range(world)}


def _p2p_collectives_worker(rank, world, port, ret):
    """`P2PCollectives.all_gather` / `reduce_scatter` land their pulls in FRESH tensors.  The caching allocator hands a block out
    again by stream order of the COMPUTE stream; the pulls run on copy streams - without the fence in `_stage` they could land while
    kernels of the block's previous owner are still queued (the failure the exchange's self-test showed once in three runs with
    five ranks, dist.py `selftest`).  Here the previous owner is made as unkind as possible: a tensor of exactly the result's size
    with a long queue of kernels writing it, freed right before the collective."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NGCF_P2P_TIMEOUT_MS="20000")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from seoul_tourism_recommendation_ngcf_amd import dist as nd
        dev = torch.device("cuda:0")
        m, d = 4096, 128
        coll = nd.P2PCollectives(None, dev, world * m * d)
        ok = True
        for it in range(6):
            send = torch.full((m, d), float(100 * it + rank), device=dev) + torch.arange(d, device=dev)
            tmp = torch.empty((world * m, d), device=dev)           # the block `full` is about to get
            for _ in range(150):
                tmp.fill_(-7.0)                                    # queued behind each other on the compute stream
            del tmp
            full = coll.all_gather(send)
            want = torch.cat([torch.full((m, d), float(100 * it + q), device=dev) + torch.arange(d, device=dev) for q in range(world)])
            ok = ok and torch.equal(full, want)
            part = torch.full((world * m, d), float(it + 1 + rank), device=dev)
            tmp = torch.empty((world, m, d), device=dev)            # the block `slots` is about to get
            for _ in range(150):
                tmp.fill_(-9.0)
            del tmp
            own = coll.reduce_scatter(part)
            ok = ok and torch.equal(own, torch.full((m, d), float(sum(it + 1 + q for q in range(world))), device=dev))
        torch.cuda.synchronize()
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_p2p_collectives_fence_their_fresh_destinations():
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_p2p_collectives_worker, args=(3, _free_port(), ret), nprocs=3, join=True)
        assert dict(ret) == {0: True, 1: True, 2: True}
