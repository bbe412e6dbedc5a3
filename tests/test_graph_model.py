"""SURVEY.md §8 row A0 and A12 against a second, independent implementation (tests/graph_model.py, plain Python).

The oracle links the product's graph builder (one restatement of src/AlignmentGraph.cpp), so oracle-vs-product comparisons say
nothing about A0. These tests do: node numbering (libstdc++ hash-map order), split nodes, neighbour order, component order, the
meaning of the MPC index by brute-force reachability, and chaining by a quadratic DP."""
import os
import random
import shutil
import subprocess

import numpy as np
import pytest

from graph_model import GraphModel, StdUnorderedOrder, chain_bruteforce
from oracle import Oracle

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


# ---- the hash-map order model against the real container --------------------------------------------------------------

PROBE = r"""
#include <cstdio>
#include <cstdlib>
#include <unordered_map>
#include <string>
struct NodePos { int id; bool end; bool operator==(const NodePos& o) const { return id == o.id && end == o.end; } };
namespace std { template <> struct hash<NodePos> { size_t operator()(const NodePos& x) const { return hash<int>()(x.id) ^ hash<bool>()(x.end); } }; }
int main(int argc, char** argv) {
    // stdin: mode (0 = int keys, 1 = NodePos keys), then keys; prints the iteration order
    int mode; if (scanf("%d", &mode) != 1) return 1;
    if (mode == 0) { std::unordered_map<int, std::string> m; int k; while (scanf("%d", &k) == 1) m[k] = "x"; for (auto& p : m) printf("%d\n", p.first); }
    else { std::unordered_map<NodePos, int> m; int k, e; while (scanf("%d %d", &k, &e) == 2) m[NodePos { k, e != 0 }] += 1; for (auto& p : m) printf("%d %d\n", p.first.id, (int)p.first.end); }
    return 0;
}
"""


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("no g++ to build the container probe")
    d = tmp_path_factory.mktemp("probe")
    src, exe = d / "probe.cpp", d / "probe"
    src.write_text(PROBE)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", str(exe), str(src)])
    return str(exe)


def test_hash_map_order_model_matches_libstdcxx(probe):
    rng = random.Random(4)
    for trial in range(12):
        n = rng.choice([1, 5, 13, 14, 29, 30, 200, 1000, 5000])
        if trial % 3 == 0:
            keys = list(range(n))                                  # GFA ids in file order
        elif trial % 3 == 1:
            keys = list(range(n)); rng.shuffle(keys)               # L lines naming segments before their S lines
        else:
            keys = [rng.randrange(4 * n + 1) for _ in range(n)]    # repeats: re-insertion keeps the position
        m = StdUnorderedOrder()
        for k in keys:
            m.insert(k)
        out = subprocess.run([probe], input="0\n" + "\n".join(map(str, keys)) + "\n", capture_output=True, text=True, check=True).stdout.split()
        assert [int(x) for x in out] == m.order(), (trial, n)
        pairs = [(rng.randrange(n + 1), rng.random() < 0.5) for _ in range(2 * n)]
        m = StdUnorderedOrder(lambda k: k[0] ^ (1 if k[1] else 0))
        for k in pairs:
            m.insert(k)
        out = subprocess.run([probe], input="1\n" + "\n".join(f"{a} {int(b)}" for a, b in pairs) + "\n", capture_output=True, text=True, check=True).stdout.split()
        got = [(int(out[i]), out[i + 1] == "1") for i in range(0, len(out), 2)]
        assert got == m.order(), (trial, n)


# ---- graphs -----------------------------------------------------------------------------------------------------------

def random_dag_gfa(rng, n_segments, minus_links=True, shuffle=True, names=False, iupac=False):
    """A DAG over segments 0..n-1 (edges low -> high), with bubbles, long segments (several split nodes), '-' orientation links
    written from the other strand, S/L lines shuffled (L lines may come before the S lines they name), non-numeric names."""
    seqs = []
    for i in range(n_segments):
        length = rng.choice([1, 1, 2, 5, 30, 64, 65, 130, 200])
        s = "".join(rng.choice("ACGT") for _ in range(length))
        if iupac and rng.random() < 0.1:
            k = rng.randrange(length)
            s = s[:k] + rng.choice("NRYKMSW") + s[k + 1:]
        seqs.append(s)
    edges = set()
    for i in range(n_segments - 1):
        edges.add((i, i + 1))                                      # a backbone keeps most of it connected
        for _ in range(rng.choice([0, 0, 1, 2])):
            j = rng.randrange(i + 1, min(n_segments, i + 6))
            edges.add((i, j))
    if n_segments > 8:                                             # a second weakly connected component
        cut = n_segments * 2 // 3
        edges = {(a, b) for a, b in edges if not (a < cut <= b)}
    name = (lambda i: f"seg{i}x") if names else (lambda i: str(i + 1))
    lines = [f"S\t{name(i)}\t{seqs[i]}" for i in range(n_segments)]
    for a, b in sorted(edges):
        if minus_links and rng.random() < 0.4:
            lines.append(f"L\t{name(b)}\t-\t{name(a)}\t-\t0M")     # the same adjacency seen from the reverse strand
        else:
            lines.append(f"L\t{name(a)}\t+\t{name(b)}\t+\t0M")
    if shuffle:
        rng.shuffle(lines)
    return "H\tVN:Z:1.0\n" + "\n".join(lines) + "\n"


def graph_cases(tmp_path):
    yield os.path.join(GOLD, "ref_test_graph.gfa")
    yield os.path.join(GOLD, "syn20k.gfa")
    from graphchainer_amd.synth import SynthGenome
    p = str(tmp_path / "genome.gfa")
    SynthGenome(3, 6000, seed=3, multi_allelic=0.3, nested=0.5, minus_links=0.4, repeats=3, repeat_len=500).write_gfa(p)   # 6 components, cover width up to 4
    yield p
    rng = random.Random(11)
    for k in range(8):
        p = str(tmp_path / f"rand{k}.gfa")
        open(p, "w").write(random_dag_gfa(rng, rng.choice([3, 9, 40, 150]), minus_links=k % 2 == 1, shuffle=k >= 2, names=k % 3 == 2, iupac=k >= 5))
        yield p


def csr(off, adj):
    return [list(adj[off[i]:off[i + 1]]) for i in range(len(off) - 1)]


def test_graph_arrays_match_the_python_model(tmp_path):
    for path in graph_cases(tmp_path):
        model = GraphModel(open(path).read())
        o = Oracle(path, long_pass=False)
        arr = {k: o.graph_array(k) for k in ("nodeLength", "nodeOffset", "nodeIDs", "reverse", "out_off", "out_adj", "in_off", "in_adj", "sequence", "componentNumber", "component_map", "component_idx")}
        assert list(arr["nodeLength"]) == model.length, path
        assert list(arr["nodeOffset"]) == model.offset, path
        assert list(arr["nodeIDs"]) == model.ids, path               # node numbering = hash-map iteration order x strand x 64 bp pieces
        assert list(arr["reverse"]) == model.reverse, path
        assert csr(arr["out_off"], arr["out_adj"]) == model.out, path   # neighbour ORDER matters (first-match rules in the backtrace and the bridge BFS)
        assert csr(arr["in_off"], arr["in_adj"]) == model.inn, path
        assert bytes(arr["sequence"].astype(np.uint8)).decode() == "".join(model.seq), path
        assert list(arr["componentNumber"]) == model.component_number(), path
        comp, idx, _ = model.weak_components()
        assert list(arr["component_map"]) == comp and list(arr["component_idx"]) == idx, path


def test_mpc_index_means_what_chaining_assumes(tmp_path):
    """src/AlignmentGraph.cpp:1328-1391. The cover: every node on some path, consecutive path nodes joined by an edge. The index:
    paths[v] = the paths through v; backwards[v] = for every path k the LAST node of k that reaches v, v itself excluded - the
    entries chaining uses to ask "which anchors end somewhere that reaches my start" (:1762-1768,1834-1845). Checked by BFS."""
    for path in graph_cases(tmp_path):
        model = GraphModel(open(path).read())
        o = Oracle(path, long_pass=False)
        g = {k: o.graph_array(k) for k in ("mpc_path_comp", "mpc_path_off", "mpc_path_nodes", "paths_off", "paths", "back_off", "back_node", "back_path", "component_map", "topo_id", "mpc_width")}
        comp = list(g["component_map"])
        n_comp = max(comp) + 1
        cover = {c: [] for c in range(n_comp)}
        for p in range(len(g["mpc_path_comp"])):
            cover[int(g["mpc_path_comp"][p])].append([int(v) for v in g["mpc_path_nodes"][g["mpc_path_off"][p]:g["mpc_path_off"][p + 1]]])
        assert [len(cover[c]) for c in range(n_comp)] == list(g["mpc_width"])
        covered = set()
        for c, plist in cover.items():
            for nodes in plist:
                assert nodes and all(comp[v] == c for v in nodes)
                assert all(b in model.out[a] for a, b in zip(nodes, nodes[1:])), path      # a path of the graph
                assert len(set(nodes)) == len(nodes)
                covered.update(nodes)
        assert covered == set(range(model.n)), path
        # topo_id is a topological order inside each component
        for v in range(model.n):
            assert all(g["topo_id"][w] > g["topo_id"][v] for w in model.out[v])
        sample = range(model.n) if model.n <= 400 else random.Random(1).sample(range(model.n), 400)
        for v in sample:
            plist = cover[comp[v]]
            on = [k for k, nodes in enumerate(plist) if v in nodes]
            assert [int(x) for x in g["paths"][g["paths_off"][v]:g["paths_off"][v + 1]]] == on, (path, v)
            anc = model.ancestors(v)
            want = []
            for k, nodes in enumerate(plist):
                reaching = [i for i, u in enumerate(nodes) if u in anc and u != v]
                if reaching:
                    want.append((nodes[max(reaching)], k))
            got = list(zip((int(x) for x in g["back_node"][g["back_off"][v]:g["back_off"][v + 1]]), (int(x) for x in g["back_path"][g["back_off"][v]:g["back_off"][v + 1]])))
            assert got == want, (path, v)


def test_chaining_against_bruteforce(tmp_path):
    """The oracle's chains (endpoint sweep over the MPC index with treaps, as the reference) against a quadratic DP over plain BFS
    reachability on the same anchors - non-overlapping fragments, overlapping ones (split_gap 18: the `x <= y_i < y` branch), and a
    small band that makes more fragments fail (sparser, gappier anchor sets)."""
    from graphchainer_amd.synth import SynthGraph
    sg = SynthGraph(60_000, seed=31)
    gfa = str(tmp_path / "g.gfa")
    sg.write_gfa(gfa)
    reads = sg.sample_reads(6, 2500, seed=5)
    reads.append(reads[0][:900] + reads[1][700:1900])          # chimeric: the best chain has to choose
    reads.append(reads[2][:600] + reads[2][1500:])             # deletion: anchors on both sides, far apart in the graph
    model = GraphModel(open(gfa).read())
    comp, _, _ = model.weak_components()
    checked = 0
    for kw in ({}, {"split_gap": 18}, {"split_gap": 18, "bandwidth": 3}, {"split_len": 50, "split_gap": 10}):
        res = Oracle(gfa, long_pass=False, **kw).align(reads)
        for r in range(len(reads)):
            a0, a1 = res["read_anchor_off"][r], res["read_anchor_off"][r + 1]
            anchors = []
            for a in range(a0, a1):
                p = [int(v) for v in res["anchor_path"][res["anchor_path_off"][a]:res["anchor_path_off"][a + 1]]]
                anchors.append((p, int(res["anchor_x"][a]), int(res["anchor_y"][a])))
            if not anchors:
                continue
            chain, score = chain_bruteforce(model.out, model.inn, comp, anchors)
            got = [int(c) for c in res["chain"][res["read_chain_off"][r]:res["read_chain_off"][r + 1]]]
            assert got == chain, (kw, r, len(anchors))
            assert int(res["chain_score"][r]) == score, (kw, r)
            checked += len(anchors)
    assert checked > 1500


def test_chaining_across_components_and_wide_covers(tmp_path):
    """Several chromosomes (anchors of one read in more than one weakly connected component: the first strictly greater score over
    increasing component id wins, src/AlignmentGraph.cpp:1722-1733), multi-allelic and nested bubbles (path-cover width > 2),
    reverse-strand links, repeats (several anchors per fragment), overlapping fragments."""
    from graphchainer_amd.synth import SynthGenome
    gen = SynthGenome(3, 40_000, seed=17, multi_allelic=0.3, nested=0.4, minus_links=0.3, repeats=6, repeat_len=1500)
    gfa = str(tmp_path / "g.gfa")
    gen.write_gfa(gfa)
    reads = gen.sample_reads(6, 3000, seed=9)
    reads.append(reads[0][:1400] + reads[1][:1400])            # two chromosomes in one read, equal halves: the tie rule decides
    reads.append(reads[2][:1000] + reads[4][:2000])
    model = GraphModel(open(gfa).read())
    comp, _, members = model.weak_components()
    assert len(members) == 6
    widths = Oracle(gfa, long_pass=False).graph_array("mpc_width")
    assert max(widths) >= 3
    multi = 0
    for kw in ({}, {"split_gap": 18}):
        res = Oracle(gfa, long_pass=False, **kw).align(reads)
        for r in range(len(reads)):
            a0, a1 = res["read_anchor_off"][r], res["read_anchor_off"][r + 1]
            anchors = []
            for a in range(a0, a1):
                p = [int(v) for v in res["anchor_path"][res["anchor_path_off"][a]:res["anchor_path_off"][a + 1]]]
                anchors.append((p, int(res["anchor_x"][a]), int(res["anchor_y"][a])))
            if not anchors:
                continue
            multi += len({comp[p[-1]] for p, _, _ in anchors}) > 1
            chain, score = chain_bruteforce(model.out, model.inn, comp, anchors)
            got = [int(c) for c in res["chain"][res["read_chain_off"][r]:res["read_chain_off"][r + 1]]]
            assert got == chain and int(res["chain_score"][r]) == score, (kw, r, len(anchors))
    assert multi >= 2
