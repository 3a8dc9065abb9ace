"""Design check on the CPU (no GPU, no product code): a sequential Python MODEL of the merged-beam kernel's traversal
(flatnav_amd/csrc/merged_beam.hpp: one sorted beam, one stable merge per link row, the tie conditions (a) eviction, (b)
selection, (d) result -- restated here line by line) against the oracle, for a planned kernel feature:

  * a query the model finishes WITHOUT a tie flag returns the oracle's ids, distances and counters (what the GPU parity tests
    show for the kernel itself; here it pins the model);
  * a query whose ONLY flag is (d) -- equal distances among the first K results -- has, by the kernel's own argument, the
    reference's traversal; so the log of its evaluated neighbours, pushed through the reference's `neighbors` heap and
    result assembly (oracle.replay_neighbors: real std::priority_queue / std::sort), must give the oracle's result ORDER.
    Today the kernel searches such a query a second time with the exact two-heap code (one whole exact-search latency: 367 us
    next to 301 us on the 1M x 128 bench index, profiles/r4_launch_timeline.md); a replay of the log touches no vector.

The model is test infrastructure like the oracle; nothing here touches visited.hpp / merged_beam.hpp themselves.
"""
import numpy as np
import pytest

from oracle import oracle as orc

INF = float("inf")


def _graph(ix, n, dim, M, dtype):
    esize = np.dtype(dtype).itemsize
    node = dim * esize + M * 4 + 4
    blob = np.frombuffer(ix.blob(), dtype=np.uint8)[: n * node].reshape(n, node)
    X = blob[:, : dim * esize].copy().view(dtype).reshape(n, dim)
    links = blob[:, dim * esize: dim * esize + M * 4].copy().view(np.uint32).reshape(n, M)
    return X, links


def _l2(q, x):
    d = q.astype(np.int64) - x.astype(np.int64)
    return np.float32((d * d).sum())


def model_search(X, links, q, K, ef, n_init=100):
    """-> (tie class 0/1/2/3, result dists, result ids, log dists, log ids, n_dist, n_hops)"""
    n = len(X)
    B = max(ef, K)
    step = max(1, n // n_init)
    best, entry = np.float32(3.4028234663852886e38), 0  # FLT_MAX
    for node in range(0, n, step):  # Index.h:845-870: first minimum wins
        d = _l2(q, X[node])
        if d < best:
            best, entry = d, node
    keys, ids, exp = [float(best)], [int(entry)], [False]
    visited = {int(entry)}
    log_d, log_i = [float(best)], [int(entry)]
    stream = []  # the hand-over log (round 5): ("H", node, evaluated) per hop, then ("C", d, id) of the row's neighbours that could still be admitted
    max_dist, amb, pend, pend_cut, tie = float(best), INF, -INF, False, 0
    sel_ever = False  # a selection tie occurred at all (even one the kernel proves harmless for the beam: the ORDER of the
                      # tied expansions, hence of the log, is then the reference's choice)
    n_dist = n_hops = 0
    while True:
        unexp = [i for i in range(len(keys)) if not exp[i]]
        if not unexp:
            break
        i1 = unexp[0]
        key_c, node = keys[i1], ids[i1]
        if len(unexp) > 1 and keys[unexp[1]] == key_c:
            pend = max(pend, key_c)
            sel_ever = True
        exp[i1] = True
        if key_c >= amb:
            tie = 1
            break
        if key_c > pend and pend > -INF:
            if pend_cut or (len(keys) >= B and not (max_dist > pend)):
                tie = 2
                break
            pend = -INF
        n_hops += 1
        new = []
        for nb in links[node]:
            nb = int(nb)
            if nb in visited:
                continue
            visited.add(nb)
            new.append(nb)
        stream.append(("H", node, len(new)))
        if not new:
            continue
        n_dist += len(new)
        d = [float(_l2(q, X[nb])) for nb in new]
        log_d += d
        log_i += new
        full0 = len(keys) >= B
        pm = [j for j in range(len(new)) if (d[j] < max_dist if full0 else True)]
        if pend > -INF and full0 and any(x == max_dist for x in d):
            pend_cut = True
        stream += [("C", d[j], new[j]) for j in pm]
        if pm:
            # the stable merge: beam entries before candidates of equal key, candidates in (key, link order)
            items = [(keys[e], 0, e, ids[e], exp[e]) for e in range(len(keys))] + [(d[j], 1, j, new[j], False) for j in pm]
            items.sort(key=lambda t: (t[0], t[1], t[2]))
            n_new = min(B, len(items))
            outside = [t[0] for t in items[n_new:]]
            keys = [t[0] for t in items[:n_new]]
            ids = [t[3] for t in items[:n_new]]
            exp = [t[4] for t in items[:n_new]]
            max_dist = keys[-1]  # Index.h:702
            if any(k == max_dist for k in outside):
                amb = max_dist
                if pend > -INF:
                    pend_cut = True
            if max_dist < amb:
                amb = INF
    nb_ = len(keys)
    if not tie and amb < INF:
        tie = 1
    if not tie and pend > -INF and (pend_cut or (nb_ >= B and not (max_dist > pend))):
        tie = 2
    cnt = min(nb_, K)
    if not tie and any(k + 1 < nb_ and keys[k] == keys[k + 1] for k in range(cnt)):
        tie = 3
    model_search.last_stream, model_search.last_entry = stream, int(entry)
    return tie, np.array(keys[:cnt], np.float32), np.array(ids[:cnt], np.int64), log_d, log_i, n_dist, n_hops, sel_ever


@pytest.mark.parametrize("dim,vmax,ef,K", [(16, 4, 40, 10), (32, 16, 52, 10), (128, 256, 52, 10), (16, 4, 150, 25)])
def test_result_ties_are_decided_by_replaying_the_log(dim, vmax, ef, K):
    rng = np.random.default_rng(dim * 1000 + ef)
    n, nq, M = 3000, 250, 16
    X = rng.integers(0, vmax, (n, dim)).astype(np.uint8)
    Q = rng.integers(0, vmax, (nq, dim)).astype(np.uint8)
    ix = orc.OracleIndex.create("l2", dim, n, M, "uint8")
    ix.add(X, 48)
    Xb, links = _graph(ix, n, dim, M, np.uint8)
    assert np.array_equal(Xb, X)
    want_d, want_l, st = ix.search(Q, K, ef, stats=True)
    by_class = {0: 0, 1: 0, 2: 0, 3: 0}
    replayable = 0
    resumed, from_log, of_hops = 0, 0, 0
    for qi in range(nq):
        tie, rd, ri, log_d, log_i, n_dist, n_hops, sel_ever = model_search(X, links, Q[qi], K, ef)
        by_class[tie] += 1
        cnt = int(st["count"][qi])
        # Round 5, the mid-flight hand-over: WHATEVER the flag and wherever the model stopped, the reference's search resumed
        # from the model's log -- both heaps replayed, stopped at the first hop where the reference would expand another node,
        # visited set rebuilt from the link rows of the hops taken, then the reference's own loop -- is the oracle's search:
        # ids, distances, order, counters.  (Queries without a flag as well: the hand-over must be right from any hop.)
        stream = model_search.last_stream
        hd = [float(r[2]) if r[0] == "H" else float(r[1]) for r in stream]
        hi = [int(r[1]) if r[0] == "H" else int(r[2]) for r in stream]
        hh = [1 if r[0] == "H" else 0 for r in stream]
        pd, pl, pnd, pnh, rep = ix.replay_search(Q[qi], K, ef, model_search.last_entry, hd, hi, hh)
        assert np.array_equal(pl, want_l[qi][:cnt]) and np.array_equal(pd, want_d[qi][:cnt]), (qi, tie, "resumed search")
        assert pnd == int(st["n_dist"][qi]) and pnh == int(st["n_hops"][qi]), (qi, tie, "resumed counters")
        if tie:
            resumed += 1
            from_log += rep
            of_hops += int(st["n_hops"][qi])
        if tie == 0:  # the model is the reference's search here: ids, distances, counters
            assert np.array_equal(ri, want_l[qi][:cnt]) and np.array_equal(rd, want_d[qi][:cnt]), (qi, tie)
            assert n_dist == int(st["n_dist"][qi]) and n_hops == int(st["n_hops"][qi])
        if tie in (0, 3):
            assert n_dist == int(st["n_dist"][qi]) and n_hops == int(st["n_hops"][qi]), (qi, tie)
        if tie in (0, 3) and not sel_ever:  # same traversal IN THE SAME ORDER: the log is the reference's evaluation sequence
            replayable += tie == 3
            assert n_dist == int(st["n_dist"][qi]) and n_hops == int(st["n_hops"][qi]), (qi, tie)
            pd, pi = orc.replay_neighbors(log_d, log_i, max(ef, K), K)
            assert np.array_equal(pi.astype(np.int64), want_l[qi][:cnt].astype(np.int64)), (qi, tie)
            assert np.array_equal(pd, want_d[qi][:cnt])
    print("dim=%d values<%d ef=%d: %d queries without a tie flag, %d eviction (a), %d selection (b), %d result-only (d) of which %d replayable"
          % (dim, vmax, ef, by_class[0], by_class[1], by_class[2], by_class[3], replayable))
    print("   hand-over: %d flagged queries resumed from their logs, all equal to the oracle; %.0f %% of their hops came from the log"
          % (resumed, 100.0 * from_log / max(of_hops, 1)))
