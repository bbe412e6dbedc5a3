"""Oracle against the committed golden vectors, plus independent brute-force models of what it computes."""
import os
import random

import numpy as np
import pytest

from oracle import Oracle

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def read_fasta(path):
    return [l.strip() for l in open(path) if l.strip() and not l.startswith(">")]


def test_reference_fixture_matches_survey_and_golden():
    """BASELINE config 1. SURVEY.md §8c recorded, from the reference's own sources run on test/graph.gfa +
    test/read.fa, one anchor (x=35, y=69, score=3, path 6,4,0) and chain [0]."""
    o = Oracle(os.path.join(GOLD, "ref_test_graph.gfa"))
    res = o.align(read_fasta(os.path.join(GOLD, "ref_test_read.fa")))
    assert list(res["anchor_x"]) == [35] and list(res["anchor_y"]) == [69]
    assert list(res["anchor_score"]) == [3]
    assert list(res["anchor_path"]) == [6, 4, 0]
    assert list(res["chain"]) == [0]
    want = np.load(os.path.join(GOLD, "ref_test.expected.npz"))
    for k in want.files:
        assert np.array_equal(res[k], want[k]), k
    g = np.load(os.path.join(GOLD, "ref_test.graph.npz"))
    for k in g.files:
        assert np.array_equal(o.graph_array(k), g[k]), k


def test_synthetic_golden():
    o = Oracle(os.path.join(GOLD, "syn20k.gfa"))
    res = o.align(read_fasta(os.path.join(GOLD, "syn20k.fa")))
    want = np.load(os.path.join(GOLD, "syn20k.expected.npz"))
    for k in want.files:
        assert np.array_equal(res[k], want[k]), k
    assert int(res["read_chain_off"][-1]) > 0 and not res["failed_assertion"].any()
    # output encoders (GAF both cigar styles, protobuf-JSON): frozen text, plus self-consistency of every GAF line
    assert o.gaf(False) == open(os.path.join(GOLD, "syn20k.expected.gaf"), "rb").read()
    assert o.gaf(True) == open(os.path.join(GOLD, "syn20k.expected.merged.gaf"), "rb").read()
    assert o.json() == open(os.path.join(GOLD, "syn20k.expected.json"), "rb").read()
    import json
    import re
    lines = o.gaf(False).decode().splitlines()
    objs = [json.loads(x) for x in o.json().decode().splitlines()]
    assert len(lines) == len(objs) >= 6
    for line, obj in zip(lines, objs):
        f = line.split("\t")
        cigar = re.findall(r"(\d+)([=XIDM])", f[-1][len("cg:Z:"):])
        on_read = sum(int(n) for n, op in cigar if op in "=XI")
        on_path = sum(int(n) for n, op in cigar if op in "=XD")
        assert on_read == int(f[3]) - int(f[2]) == len(obj["sequence"])            # read span = aligned sequence
        assert on_path == int(f[8]) - int(f[7])                                      # path span
        assert int(f[9]) == sum(int(n) for n, op in cigar if op == "=")             # matches
        assert len(re.findall(r"[<>]", f[5])) == len(obj["path"]["mapping"])         # one mapping per path step
        edits = [e for m in obj["path"]["mapping"] for e in m["edit"]]
        assert sum(e.get("to_length", 0) for e in edits) == on_read and sum(e.get("from_length", 0) for e in edits) == on_path


def test_whole_read_assertion_leaves_the_read_with_nothing(monkeypatch, tmp_path):
    """src/Aligner.cpp:529,591,702: `cont` is declared once per read; after the whole-read pass has thrown, the fragment loop
    still runs but keeps no anchor, so the read ends with no anchors, no chain and no alignment. The other reads of the run are
    unaffected (same arrays as without the hook)."""
    from graphchainer_amd.synth import SynthGraph
    sg = SynthGraph(40_000, seed=5)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(3, 2000, seed=3)
    base = Oracle(gfa, long_pass=True).align(reads)
    monkeypatch.setenv("GC_TEST_FAIL_LONG", "1")
    hook = Oracle(gfa, long_pass=True).align(reads)
    assert list(hook["failed_assertion"]) == [0, 1, 0]
    for off, keys in (("read_anchor_off", ["anchor_x", "anchor_score", "anchor_first_node"]), ("read_chain_off", ["chain"]),
                      ("read_longall_off", ["longall_start", "longall_end", "longall_score"]), ("read_path_off", ["path_node", "path_offset"])):
        assert hook[off][2] == hook[off][1], off
        assert base[off][2] > base[off][1], off
        for k in keys:
            kept = np.concatenate([base[k][:base[off][1]], base[k][base[off][2]:]])
            assert np.array_equal(hook[k], kept), k
    assert hook["chain_edit_distance"][1] == -1 and hook["long_edit_distance"][1] == -1 and hook["chained_better"][1] == 0


# ---- graph structure properties ---------------------------------------------------------------------

@pytest.fixture(scope="module")
def syn():
    o = Oracle(os.path.join(GOLD, "syn20k.gfa"))
    arrays = {k: o.graph_array(k) for k in ["nodeLength", "nodeIDs", "nodeOffset", "reverse", "componentNumber", "chainNumber", "out_off", "out_adj", "in_off", "in_adj", "sequence", "component_map", "mpc_width"]}
    return o, arrays


def test_component_number_is_a_topological_rank(syn):
    _, a = syn
    n = len(a["nodeLength"])
    assert sorted(a["componentNumber"]) == list(range(n))        # a DAG: every node is its own component
    for v in range(n):
        for e in range(a["out_off"][v], a["out_off"][v + 1]):
            assert a["componentNumber"][a["out_adj"][e]] > a["componentNumber"][v]


def test_in_and_out_adjacency_are_transposes(syn):
    _, a = syn
    n = len(a["nodeLength"])
    outs = {(v, int(a["out_adj"][e])) for v in range(n) for e in range(a["out_off"][v], a["out_off"][v + 1])}
    ins = {(int(a["in_adj"][e]), v) for v in range(n) for e in range(a["in_off"][v], a["in_off"][v + 1])}
    assert outs == ins


def test_strands_are_reverse_complements(syn):
    """Bigraph node 2n+1 is the reverse complement of 2n (src/BigraphToDigraph.cpp:101-104)."""
    _, a = syn
    seq = {}
    pos = 0
    order = np.lexsort((a["nodeOffset"], a["nodeIDs"]))
    starts = np.concatenate([[0], np.cumsum(a["nodeLength"])])
    for i in order:
        seq.setdefault(int(a["nodeIDs"][i]), []).append(bytes(a["sequence"][starts[i]:starts[i + 1]].astype(np.uint8)))
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for nid, parts in seq.items():
        if nid % 2 == 0:
            fw = b"".join(parts)
            bw = b"".join(seq[nid + 1])
            assert bw == fw.translate(comp)[::-1]


def test_chains_are_weakly_connected_components(syn):
    _, a = syn
    assert len(set(a["chainNumber"])) == len(set(a["component_map"])) == 2      # one per strand
    assert len(set(zip(a["chainNumber"], a["component_map"]))) == 2
    assert list(a["mpc_width"]) == [2, 2]                                       # SNP/indel bubbles: width 2


# ---- brute-force model of the fragment extension ---------------------------------------------------

def brute_force_best_alignment(arrays, start_node, start_off, seq):
    """Unbanded DP from the cell after (start_node, start_off): min edit distance of aligning all of `seq` to a
    path that starts right after the seed cell, ending anywhere (what the banded bit-vector DP approximates)."""
    n = len(arrays["nodeLength"])
    starts = np.concatenate([[0], np.cumsum(arrays["nodeLength"])])
    order = np.argsort(arrays["componentNumber"])
    INF = 10 ** 6
    L = len(seq)
    # score[v][c][r]: best cost with read rows 0..r consumed, ending at column c of node v; row -1 = nothing consumed
    col = {}
    def column(v, c):
        return col.get((v, c))
    best_end = INF
    for v in order:
        v = int(v)
        ln = int(arrays["nodeLength"][v])
        for c in range(ln):
            preds = []
            if c > 0:
                if column(v, c - 1) is not None:
                    preds.append(column(v, c - 1))
            else:
                for e in range(arrays["in_off"][v], arrays["in_off"][v + 1]):
                    u = int(arrays["in_adj"][e])
                    p = column(u, int(arrays["nodeLength"][u]) - 1)
                    if p is not None:
                        preds.append(p)
            is_seed = (v == start_node and c == start_off)
            if not preds and not is_seed:
                continue
            if is_seed:
                cur = [0] + [INF] * L       # the seed cell itself: row -1 cost 0, it consumes no read base
                col[(v, c)] = cur
                continue
            ch = chr(int(arrays["sequence"][starts[v] + c]))
            cur = [INF] * (L + 1)
            cur[0] = min(p[0] for p in preds) + 1
            for r in range(1, L + 1):
                m = cur[r - 1] + 1
                for p in preds:
                    m = min(m, p[r] + 1, p[r - 1] + (0 if seq[r - 1] == ch else 1))
                cur[r] = m
            if min(cur) > L + 12:
                continue
            col[(v, c)] = cur
            best_end = min(best_end, cur[L])
    return min(best_end, L)   # aligning everything as insertions right after the seed is always possible


def test_fragment_scores_match_unbanded_dp(syn):
    """Anchor score = forward + backward extension score; each must equal the unbanded optimum whenever the
    optimum stays inside the band (it does for these low-error fragments) and is never below it."""
    o, a = syn
    reads = read_fasta(os.path.join(GOLD, "syn20k.fa"))[:2]
    res = o.align(reads)
    rng = random.Random(0)
    n_anchor = int(res["read_anchor_off"][-1])
    picks = rng.sample(range(n_anchor), 12)
    read_of = np.searchsorted(res["read_anchor_off"], np.arange(n_anchor), side="right") - 1
    exact = 0
    for ai in picks:
        read = reads[read_of[ai]]
        x = int(res["anchor_x"][ai])
        frag = read[x:x + 35]
        # recover the seed of this anchor: the trace passes through it; use the forward part from the first cell instead:
        # align frag[1:] starting after the first trace cell, which is how a seed at p=0 would extend
        t0 = int(res["anchor_trace_off"][ai])
        node_b, off_b = int(res["anchor_trace_node"][t0]), int(res["anchor_trace_offset"][t0])
        first_seqpos = int(res["anchor_trace_seqpos"][t0])
        if first_seqpos != 0:
            continue
        # bigraph coords -> split node
        cand = [i for i in range(len(a["nodeIDs"])) if a["nodeIDs"][i] == node_b and a["nodeOffset"][i] <= off_b < a["nodeOffset"][i] + a["nodeLength"][i]]
        v = cand[0]
        c = off_b - int(a["nodeOffset"][v])
        starts = np.concatenate([[0], np.cumsum(a["nodeLength"])])
        first_match = 0 if chr(int(a["sequence"][starts[v] + c])) == frag[0].upper() else 1
        opt = brute_force_best_alignment(a, v, c, frag[1:].upper()) + first_match
        score = int(res["anchor_score"][ai])
        assert score >= opt - first_match - 1
        if score == opt:
            exact += 1
    assert exact >= 6


# ---- chain stitching against an independent model ---------------------------------------------------------

def _bridge(out_off, out_adj, node_length, start, target, sep_limit):
    """AlignmentGraph::getChainPath (src/AlignmentGraph.cpp:1866-1916) in plain Python: fewest-hops path by BFS, nodes further
    than sep_limit bp are not expanded; the distance is compared as an unsigned 64-bit number, so a negative limit is none."""
    limit = sep_limit % (1 << 64)
    queue, dist, pre = [start], {start: 0}, {}
    i = 0
    while target not in dist and i < len(queue):
        s = queue[i]
        i += 1
        if dist[s] > limit:
            continue
        for t in out_adj[out_off[s]:out_off[s + 1]]:
            t = int(t)
            if t not in dist:
                dist[t] = dist[s] + int(node_length[t])
                pre[t] = s
                queue.append(t)
    if target not in dist:
        return []
    path = [target]
    while path[-1] != start:
        path.append(pre[path[-1]])
    return path[::-1]


def _stitch(arrays, anchors, chain, colinear_gap):
    """The stitching loop of src/Aligner.cpp:754-822 + the length of pathToTrace (:409-424); returns the cells (node, offset)
    of the longest piece."""
    node_length, out_off, out_adj = arrays["nodeLength"], arrays["out_off"], arrays["out_adj"]
    best, pos_path, nodes, first_off, last_off = [], [], set(), 0, 0

    def cells_of(path, first, last):
        out = []
        for node in path:
            s, l = 0, int(node_length[node])
            if node == path[0]:
                s = first
            elif node == path[-1]:
                l = last + 1
            out.extend((node, o) for o in range(s, l))
        return out

    def keep():
        nonlocal best
        c = cells_of(pos_path, first_off, last_off)
        if len(best) < len(c):
            best = c

    for index in chain:
        a = anchors[index]
        if not pos_path:
            pos_path = list(a["path"])
            first_off, last_off = a["first_offset"], a["last_offset"]
            nodes = set(pos_path)
            continue
        gap = a["path"][0] == pos_path[-1] and colinear_gap != -1 and a["first_offset"] - last_off > colinear_gap + 1
        bridge = []
        if a["path"][0] not in nodes and pos_path[-1] != a["first_node"]:
            limit = colinear_gap
            if limit != -1:
                limit -= a["first_offset"] + (int(node_length[pos_path[-1]]) - last_off - 1)
            bridge = _bridge(out_off, out_adj, node_length, pos_path[-1], a["first_node"], limit)
            if not bridge:
                gap = True
        if gap:
            keep()
            nodes, pos_path = set(), []
            first_off = a["first_offset"]
        else:
            for j in bridge:
                if j not in nodes:
                    nodes.add(j)
                    pos_path.append(j)
        for j in a["path"]:
            if j not in nodes:
                nodes.add(j)
                pos_path.append(j)
        last_off = a["last_offset"]
    if pos_path:
        keep()
    return best


@pytest.mark.parametrize("colinear_gap", [10000, 120, -1])
def test_stitched_paths_match_independent_model(colinear_gap):
    """The oracle's stitched path of every read (reads of the golden set plus chimeras of them, whose chains break) against a
    plain-Python restatement of the loop and of the bridge search, written from the same reference lines."""
    gfa = os.path.join(GOLD, "syn20k.gfa")
    reads = read_fasta(os.path.join(GOLD, "syn20k.fa"))
    reads += [reads[0][:900] + reads[1][400:1500], reads[2][:700] + reads[3][1200:2000] + reads[4][:600]]
    o = Oracle(gfa, colinear_gap=colinear_gap, long_pass=False)
    res = o.align(reads)
    arrays = {k: o.graph_array(k) for k in ["nodeLength", "out_off", "out_adj"]}
    pieces = 0
    for r in range(len(reads)):
        a0, a1 = int(res["read_anchor_off"][r]), int(res["read_anchor_off"][r + 1])
        anchors = []
        for a in range(a0, a1):
            p0, p1 = int(res["anchor_path_off"][a]), int(res["anchor_path_off"][a + 1])
            anchors.append({"path": [int(x) for x in res["anchor_path"][p0:p1]], "first_node": int(res["anchor_first_node"][a]),
                            "first_offset": int(res["anchor_first_offset"][a]), "last_offset": int(res["anchor_last_offset"][a])})
        chain = [int(x) for x in res["chain"][int(res["read_chain_off"][r]):int(res["read_chain_off"][r + 1])]]
        want = _stitch(arrays, anchors, chain, colinear_gap)
        c0, c1 = int(res["read_path_off"][r]), int(res["read_path_off"][r + 1])
        got = list(zip((int(x) for x in res["path_node"][c0:c1]), (int(x) for x in res["path_offset"][c0:c1])))
        assert got == want, f"read {r}: {len(got)} cells, model {len(want)}"
        pieces += bool(want)
    assert pieces >= 6


def test_bridge_search_rank_pruning_changes_nothing(syn):
    """k_stitch leaves nodes whose componentNumber exceeds the target's out of its bridge search (gc_stitch.hip). That must not
    change any result: same path (or same failure) as the plain search, for targets ahead, behind, on the other strand, with
    and without a budget."""
    _, arrays = syn
    node_length, out_off, out_adj, rank = arrays["nodeLength"], arrays["out_off"], arrays["out_adj"], arrays["componentNumber"]

    def pruned(start, target, sep_limit):
        limit = sep_limit % (1 << 64)
        if rank[start] > rank[target]:
            return []
        queue, dist, pre = [start], {start: 0}, {}
        i = 0
        while target not in dist and i < len(queue):
            s = queue[i]
            i += 1
            if dist[s] > limit:
                continue
            for t in out_adj[out_off[s]:out_off[s + 1]]:
                t = int(t)
                if rank[t] > rank[target] or t in dist:
                    continue
                dist[t] = dist[s] + int(node_length[t])
                pre[t] = s
                queue.append(t)
        if target not in dist:
            return []
        path = [target]
        while path[-1] != start:
            path.append(pre[path[-1]])
        return path[::-1]

    rng = random.Random(5)
    n = len(node_length)
    by_rank = sorted(range(n), key=lambda v: int(rank[v]))
    found = 0
    for trial in range(400):
        s = rng.randrange(n)
        if trial % 2:
            t = rng.randrange(n)                                   # anywhere: mostly unreachable
        else:
            at = min(n - 1, max(0, by_rank.index(s) + rng.randrange(-20, 120)))
            t = by_rank[at]                                        # near in topological order: often reachable
        if s == t:
            continue
        for limit in (rng.randrange(0, 3000), -1, -7):
            want = _bridge(out_off, out_adj, node_length, s, t, limit)
            assert pruned(s, t, limit) == want, (s, t, limit)
            found += bool(want)
    assert found > 50


def test_minimizer_index_properties(syn):
    """The minimizer index against its definition rather than against another sliding-window implementation (k = 15, w = 20,
    src/MinimizerSeeder.cpp:104-189): (1) every indexed (k-mer, position) is the k-mer that ends there in its node, (2) it has the
    smallest hash of some window of w - k + 1 consecutive k-mers that contains it (w - k + 2 for a node's first window, as in the
    reference), (3) every window of w - k + 2 consecutive k-mers holds at least one indexed position - the guarantee seeding
    relies on, (4) k-mers are sorted and distinct and the position lists partition the positions."""
    o, arrays = syn
    k, w = 15, 20
    per_window = w - k + 1
    kmers, start, positions = (o.graph_array(n).astype(np.uint64) for n in ("index_kmers", "index_start", "index_positions"))
    assert np.all(kmers[1:] > kmers[:-1]) and start[0] == 0 and start[-1] == len(positions) and np.all(np.diff(start.astype(np.int64)) > 0)
    node_ids, node_offset, node_length = arrays["nodeIDs"], arrays["nodeOffset"], arrays["nodeLength"]
    seq_off = np.concatenate([[0], np.cumsum(node_length)])
    letters = arrays["sequence"]
    # original (bigraph) node sequences from their split nodes
    originals = {}
    for v in np.argsort(node_ids * (1 << 20) + node_offset, kind="stable"):
        originals.setdefault(int(node_ids[v]), []).append(bytes(int(c) for c in letters[seq_off[v]:seq_off[v + 1]]))
    originals = {i: b"".join(parts) for i, parts in originals.items()}
    code = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3}
    hashed = {}   # node id -> list over end positions of (kmer, hash) or None
    for i, s in originals.items():
        row = [None] * len(s)
        for end in range(k - 1, len(s)):
            word = s[end - k + 1:end + 1]
            if all(c in code for c in word):
                value = 0
                for c in word:
                    value = (value << 2) | code[c]
                row[end] = (value, int(o.lib.gco_minimizer_hash(value)))
        hashed[i] = row
    indexed = {i: set() for i in originals}
    for ki in range(len(kmers)):
        for p in positions[int(start[ki]):int(start[ki + 1])]:
            split, off = int(p) >> 6, int(p) & 63
            i, end = int(node_ids[split]), int(node_offset[split]) + off
            assert hashed[i][end] is not None and hashed[i][end][0] == int(kmers[ki])            # (1)
            indexed[i].add(end)
    checked_windows = 0
    for i, row in hashed.items():
        for end in indexed[i]:                                                                   # (2)
            h = row[end][1]
            ok = False
            for size in (per_window, per_window + 1):
                for first in range(end - size + 1, end + 1):
                    window = row[max(first, 0):first + size] if first >= 0 else []
                    if len(window) == size and all(x is not None for x in window) and min(x[1] for x in window) == h:
                        ok = True
            assert ok, (i, end)
        for first in range(k - 1, len(row) - per_window):                                      # (3)
            window = row[first:first + per_window + 1]
            if all(x is not None for x in window):
                assert any(e in indexed[i] for e in range(first, first + per_window + 1)), (i, first)
                checked_windows += 1
    assert checked_windows > 10_000 and sum(len(v) for v in indexed.values()) == len(positions)  # (4)


def test_whole_read_traces_are_walks_in_the_graph():
    """Every whole-read alignment trace, read as the reference's output coordinates (bigraph node id, offset in the original
    node, read position): consecutive cells stay put, advance one base inside the node, or cross a graph edge from a node's
    last base to the next node's first; read positions advance by 0 or 1 and never both stand still; the trace spans exactly
    the alignment's read interval; and its unit-cost edit count is the NM the GAF line reports and at least the DP score."""
    import re
    gfa = os.path.join(GOLD, "syn20k.gfa")
    reads = read_fasta(os.path.join(GOLD, "syn20k.fa"))
    reads.append(reads[0][:900] + reads[1][400:1500])
    o = Oracle(gfa, long_pass=True)
    res = o.align(reads)
    arrays = {k: o.graph_array(k) for k in ["nodeLength", "nodeIDs", "nodeOffset", "sequence", "out_off", "out_adj"]}
    seq_off = np.concatenate([[0], np.cumsum(arrays["nodeLength"])])
    original, last_split = {}, {}
    for v in np.argsort(arrays["nodeIDs"] * (1 << 20) + arrays["nodeOffset"], kind="stable"):
        i = int(arrays["nodeIDs"][v])
        original[i] = original.get(i, "") + "".join(chr(c) for c in arrays["sequence"][seq_off[v]:seq_off[v + 1]])
        last_split[i] = int(v)
    edges = set()
    for i, v in last_split.items():
        for t in arrays["out_adj"][arrays["out_off"][v]:arrays["out_off"][v + 1]]:
            if int(arrays["nodeIDs"][t]) != i:
                edges.add((i, int(arrays["nodeIDs"][t])))
    off, per_read = res["long_trace_off"], res["read_longall_off"]
    costs = {}
    for r, read in enumerate(reads):
        for a in range(int(per_read[r]), int(per_read[r + 1])):
            n, f, s = (res[k][off[a]:off[a + 1]] for k in ("long_trace_node", "long_trace_offset", "long_trace_seqpos"))
            assert int(s[0]) == int(res["longall_start"][a]) and int(s[-1]) == int(res["longall_end"][a]) - 1
            cost = int(original[int(n[0])][f[0]] != read[s[0]])
            for i in range(1, len(n)):
                same_place = n[i] == n[i - 1] and f[i] == f[i - 1]
                read_step = int(s[i]) - int(s[i - 1])
                assert read_step in (0, 1) and not (same_place and read_step == 0)
                if not same_place:
                    if n[i] == n[i - 1]:
                        assert f[i] == f[i - 1] + 1
                    else:
                        assert f[i] == 0 and f[i - 1] == len(original[int(n[i - 1])]) - 1 and (int(n[i - 1]), int(n[i])) in edges
                cost += 1 if (same_place or read_step == 0) else int(original[int(n[i])][f[i]] != read[s[i]])
            assert cost >= int(res["longall_score"][a])
            costs.setdefault((r, int(res["longall_start"][a]), int(res["longall_end"][a])), []).append(cost)
    assert sum(len(v) for v in costs.values()) >= 8
    # the GAF lines of the selected alignments report the same edit counts
    seen = 0
    for line in o.gaf(False).decode().splitlines():
        fields = line.split("\t")
        nm = int(next(x for x in fields if x.startswith("NM:i:"))[5:])
        key = (int(fields[0][1:]), int(fields[2]), int(fields[3]))
        if key in costs:
            seen += 1
            assert nm in costs[key]   # several alignments of a read can share an interval
    assert seen >= 6


# ---- GAM pinned to the reference's own schema (r5) -------------------------------------------------------------------------------------------

def test_hand_built_vg_descriptor_is_a_subset_of_the_reference_schema():
    """tests/vg_descriptor.py (what decodes GAM on the GPU box) against the field table make_gam_golden.py took from /root/reference/scripts/vg_pb2.py:
    same names, numbers, types, repeated flags and message types for every field it declares."""
    import json
    from vg_descriptor import FIELDS, TYPE_NUMBER
    schema = json.load(open(os.path.join(GOLD, "vg_schema.expected.json")))
    assert schema["package"] == "vg" and set(FIELDS) == set(schema["messages"])
    for message, fields in FIELDS.items():
        by_name = {f["name"]: f for f in schema["messages"][message]}
        for name, number, ftype, repeated, type_name in fields:
            ref = by_name[name]
            assert (ref["number"], ref["type"], ref["repeated"], ref["message"]) == (number, TYPE_NUMBER[ftype], repeated, type_name), (message, name)
    # and the path sets nothing the subset lacks: every fixture message decoded with the FULL descriptor uses only these fields (checked by the generator through the
    # names round trip; here: the keys of the committed documents)
    allowed = {m: {f[0] for f in fs} for m, fs in FIELDS.items()}
    for case in ("ref_test", "syn20k", "syn20k_more"):
        for group in json.load(open(os.path.join(GOLD, case + ".expected.gam.json")))["groups"]:
            for aln in group:
                assert set(aln) <= allowed["Alignment"] and set(aln["path"]) <= allowed["Path"]
                for m in aln["path"]["mapping"]:
                    assert set(m) <= allowed["Mapping"] and set(m["position"]) <= allowed["Position"]
                    assert all(set(e) <= allowed["Edit"] for e in m["edit"])


@pytest.mark.parametrize("case", ["ref_test", "syn20k", "syn20k_more"])
def test_oracle_gam_equals_the_reference_decoded_fixture(case):
    """The oracle's GAM stream (oracle/output.hpp: vgAlignmentToProto, gamGroup) byte for byte and message for message against the fixture that the reference's own
    vg_pb2 descriptor and summary.py reader decoded (tests/golden/make_gam_golden.py; the generator required every message to re-serialise to itself)."""
    from vg_descriptor import decode_gam_stream, golden_case
    gfa, reads, want_groups, want_stream, want_better = golden_case(case)
    ora = Oracle(gfa, long_pass=True)
    res = ora.align(reads)
    groups = ora.gam_groups()
    assert b"".join(groups) == want_stream
    assert decode_gam_stream(b"".join(groups)) == want_groups
    assert [int(x) for x in res["chained_better"]] == want_better
    assert sum(len(g) for g in want_groups) == len(ora.json().decode().splitlines())
    if case == "syn20k_more":
        assert sum(want_better) >= 1 and len(want_groups) == len(reads) - 1          # a chained winner is in the fixture; the ten-base read has no group
        assert any(m["position"].get("is_reverse") for g in want_groups for a in g for m in a["path"]["mapping"])


@pytest.mark.parametrize("case", ["ref_test", "syn20k", "syn20k_more"])
def test_oracle_paths_spelled_through_the_gfa_give_the_reported_distances(case):
    """The CONTENT of the output read back the way the reference's own harness reads it (scripts/summary.py:77-91, restated next to the reference's descriptor in
    tests/golden/make_gam_golden.py, which committed what it saw): every alignment's path spelled through the GFA by node_id / is_reverse equals the fixture's, and the
    part of it the alignment covers has the NW edit distance to the read that the pipeline REPORTS (long_edit_distance of the selection's first alignment,
    src/Aligner.cpp:376-408; chain_edit_distance of the chained one, :845) - the harness's global_ed_read_long / global_ed_read_clcs columns on the covered part."""
    from vg_descriptor import decode_gam_stream, golden_case, golden_paths, load_gfa_segments, spell_alignment
    from oracle.binding import load_oracle_lib
    gfa, reads, _, _, want_better = golden_case(case)
    ora = Oracle(gfa, long_pass=True)
    res = ora.align(reads)
    groups = decode_gam_stream(b"".join(ora.gam_groups()))
    want = golden_paths(case)
    VL = load_gfa_segments(gfa)
    lib = load_oracle_lib()
    assert len(groups) == len(want)
    for group, row in zip(groups, want):
        r = row["read"]
        distances = []
        for aln, want_aln in zip(group, row["alignments"]):
            got, part = spell_alignment(aln, VL)
            assert got == {k: want_aln[k] for k in got}, (case, r)
            distances.append(int(lib.gco_edit_distance(part.encode(), len(part), reads[r], len(reads[r]))))
        assert distances == [a["nw_distance_to_read"] for a in row["alignments"]]
        reported = int(res["chain_edit_distance"][r]) if res["chained_better"][r] else int(res["long_edit_distance"][r])
        assert reported == row["reported_distance"] and reported in distances and (len(distances) > 1 or distances[0] == reported)
    assert sum(len(row["alignments"]) for row in want) >= len(want) >= 1


# ---- the one DEFINED rule, measured (r5): flattenLastSliceEnd's tie order ---------------------------------------------------------------------

def _tie_sensitivity(gfa, reads, **kw):
    """Aligns `reads` with the defined tie order (band-entry order) and with its reverse; per read: did anchors / chain / stitched path / decision / GAF lines move?"""
    per_order = []
    for order in (0, 1):
        o = Oracle(gfa, long_pass=True, tie_order=order, **kw)
        res = o.align(reads)
        lines = o.gaf(False).split(b"\n")[:-1]
        per_read_lines = {}
        for line in lines:
            per_read_lines.setdefault(int(line.split(b"\t", 1)[0][1:]), []).append(line)
        per_order.append((res, per_read_lines))
    (a, la), (b, lb) = per_order
    n = len(reads)
    seg = lambda res, off, keys, r: tuple(tuple(res[k][res[off][r]:res[off][r + 1]].tolist()) for k in keys)
    moved = {"anchors": 0, "chain": 0, "path": 0, "whole_read_alignments": 0, "decision_or_distances": 0, "gaf_lines": 0}
    moved_reads = set()
    for r in range(n):
        d = {
            "anchors": seg(a, "read_anchor_off", ("anchor_x", "anchor_score", "anchor_first_node", "anchor_first_offset", "anchor_last_node", "anchor_last_offset"), r)
                       != seg(b, "read_anchor_off", ("anchor_x", "anchor_score", "anchor_first_node", "anchor_first_offset", "anchor_last_node", "anchor_last_offset"), r),
            "chain": seg(a, "read_chain_off", ("chain",), r) != seg(b, "read_chain_off", ("chain",), r) or a["chain_score"][r] != b["chain_score"][r],
            "path": seg(a, "read_path_off", ("path_node", "path_offset"), r) != seg(b, "read_path_off", ("path_node", "path_offset"), r),
            "whole_read_alignments": seg(a, "read_longall_off", ("longall_start", "longall_end", "longall_score"), r) != seg(b, "read_longall_off", ("longall_start", "longall_end", "longall_score"), r),
            "decision_or_distances": (a["chained_better"][r], a["long_edit_distance"][r], a["chain_edit_distance"][r]) != (b["chained_better"][r], b["long_edit_distance"][r], b["chain_edit_distance"][r]),
            "gaf_lines": la.get(r, []) != lb.get(r, []),
        }
        for k, v in d.items():
            moved[k] += bool(v)
        if any(d.values()):
            moved_reads.add(r)
    tied = {r for r in range(n) if a["flatten_ties"][r] + a["flatten_ties_long"][r] > 0}
    return {"reads": n, "flatten_calls": int(a["flatten_counters"][0]), "tied_extensions": int(a["flatten_counters"][1]), "tied_reads": len(tied),
            "tied_reads_whole_read_pass": int((a["flatten_ties_long"] > 0).sum()), "moved": moved, "moved_reads": len(moved_reads)}, tied, moved_reads


def test_tie_order_sensitivity_is_counted_and_confined_to_tied_reads(tmp_path, capsys):
    """SURVEY.md §8(c) / VERDICT r4: flattenLastSliceEnd (src/GraphAlignerBitvectorCommon.h:1170-1229) takes its minimum with a strict '<' in the iteration order of a
    parallel-hashmap (src/NodeSlice.h:54) this tree does not hold; the build defines band-entry order. The oracle counts the extensions whose backtrace started from a
    minimum attained in more than one node (flatten_ties / flatten_ties_long per read: the same arrays the product returns) and can run with the order REVERSED. Checked:
    a read without a tie gives the same answer under both orders (so `flatten_ties == 0` is a certificate), and the numbers DESIGN.md §7 quotes are what this prints."""
    from graphchainer_amd.synth import SynthGraph
    report = {}
    syn = read_fasta(os.path.join(GOLD, "syn20k.fa"))
    report["syn20k"], tied, moved = _tie_sensitivity(os.path.join(GOLD, "syn20k.gfa"), syn)
    assert moved <= tied
    sg = SynthGraph(300_000, seed=7)
    gfa = str(tmp_path / "g300k.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(48, 10_000, seed=11)
    report["300 kbp, 48 x 10 kb"], tied, moved = _tie_sensitivity(gfa, reads)
    assert moved <= tied, sorted(moved - tied)                       # only reads that had a tie can depend on the order
    assert report["300 kbp, 48 x 10 kb"]["tied_extensions"] > 0     # the case is exercised (SURVEY: about one tie per 10 kb read)
    reads = sg.sample_reads(200, 1500, seed=12)
    report["300 kbp, 200 x 1.5 kb"], tied, moved = _tie_sensitivity(gfa, reads)
    assert moved <= tied, sorted(moved - tied)
    import json
    with capsys.disabled():
        print("\n[tie order sensitivity] " + json.dumps(report))
