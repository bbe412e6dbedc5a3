"""Independent plain-Python models of the start-up data and of co-linear chaining (row A12).

Nothing here shares code with the product or the oracle: it restates, from the reference's sources, what the arrays must be.

  * libstdc++'s unordered_map iteration order (the reference's node numbering IS that order: `for (auto node : graph.nodes)`,
    src/BigraphToDigraph.cpp:229, and its neighbour order comes from `for (auto edge : graph.edges)`, :251). The model follows
    GCC's _Hashtable: insertion at the head of the bucket / of the list, prime bucket counts from libstdc++'s own prime table,
    rehash relinking nodes in list order. tests/test_graph_model.py checks it against a probe compiled with the local g++.
  * GFA -> bigraph nodes (2n forward, 2n+1 reverse complement) -> split nodes of <= 64 bp -> edges -> ambiguous nodes moved to
    the end (src/GfaGraph.cpp:212-290, src/BigraphToDigraph.cpp:101-132,215-267, src/AlignmentGraph.cpp:51-253,919-974),
    the Tarjan component order on a DAG (:1008-1115) and the weakly connected components (:1430-1463).
  * brute-force meaning of the MPC index (:1328-1391) and a quadratic chaining DP over BFS reachability (:1712-1863).
"""
import ctypes
from collections import deque

# ---------------------------------------------------------------------------------------------- libstdc++ hash table order


def _prime_list():
    lib = ctypes.CDLL("libstdc++.so.6")
    arr = (ctypes.c_ulong * 305).in_dll(lib, "_ZNSt8__detail12__prime_listE")   # std::__detail::__prime_list: 256 + 48 primes + sentinel
    return list(arr)


_FAST_BKT = [2, 2, 2, 3, 5, 5, 7, 7, 11, 11, 11, 11, 13, 13]


class StdUnorderedOrder:
    """Iteration order of a libstdc++ std::unordered_map / unordered_set with unique keys (max_load_factor 1)."""

    def __init__(self, hash_fn=lambda k: k):
        self.hash = hash_fn
        self.primes = _prime_list()
        self.n_bkt = 1
        self.next_resize = 0
        self.keys = set()
        # singly linked list: self.next[key] = following key (None at the end); "before begin" is the sentinel object BB
        self.BB = object()
        self.next = {self.BB: None}
        self.buckets = {}          # bucket index -> the node BEFORE the bucket's first node

    def _next_bkt(self, n):
        if n < len(_FAST_BKT):
            if n == 0:
                return 1
            self.next_resize = _FAST_BKT[n]
            return _FAST_BKT[n]
        n_primes = len(self.primes) - 1
        last = n_primes - 1
        lo, hi = 6, last             # std::lower_bound(__prime_list + 6, __last_prime, n)
        while lo < hi:
            mid = (lo + hi) // 2
            if self.primes[mid] < n:
                lo = mid + 1
            else:
                hi = mid
        self.next_resize = (1 << 64) - 1 if lo == last else self.primes[lo]
        return self.primes[lo]

    def _need_rehash(self, n_elt, n_ins):
        if n_elt + n_ins > self.next_resize:
            min_bkts = max(n_elt + n_ins, 0 if self.next_resize else 11)
            if min_bkts >= self.n_bkt:
                return self._next_bkt(max(min_bkts + 1, self.n_bkt * 2))
            self.next_resize = self.n_bkt
        return None

    def _bkt(self, key, n=None):
        return (self.hash(key) & ((1 << 64) - 1)) % (n or self.n_bkt)

    def _rehash(self, n):
        new_buckets = {}
        p = self.next[self.BB]
        self.next[self.BB] = None
        bbegin_bkt = 0
        while p is not None:
            nxt = self.next[p]
            b = self._bkt(p, n)
            if b not in new_buckets:
                self.next[p] = self.next[self.BB]
                self.next[self.BB] = p
                new_buckets[b] = self.BB
                if self.next[p] is not None:
                    new_buckets[bbegin_bkt] = p
                bbegin_bkt = b
            else:
                before = new_buckets[b]
                self.next[p] = self.next[before]
                self.next[before] = p
            p = nxt
        self.buckets = new_buckets
        self.n_bkt = n

    def insert(self, key):
        if key in self.keys:
            return False
        n = self._need_rehash(len(self.keys), 1)
        if n is not None:
            self._rehash(n)
        b = self._bkt(key)
        if b in self.buckets:
            before = self.buckets[b]
            self.next[key] = self.next[before]
            self.next[before] = key
        else:
            self.next[key] = self.next[self.BB]
            self.next[self.BB] = key
            if self.next[key] is not None:
                self.buckets[self._bkt(self.next[key])] = key
            self.buckets[b] = self.BB
        self.keys.add(key)
        return True

    def order(self):
        out, p = [], self.next[self.BB]
        while p is not None:
            out.append(p)
            p = self.next[p]
        return out


# ---------------------------------------------------------------------------------------------- GFA -> alignment graph

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N", "U": "A", "R": "Y", "Y": "R", "K": "M", "M": "K", "S": "S", "W": "W", "B": "V", "V": "B", "D": "H", "H": "D"}


def revcomp(s):
    return "".join(_COMP[c.upper()] for c in reversed(s))


class GraphModel:
    def __init__(self, gfa_text):
        names = {}

        def name_id(nm):                      # getNameId, src/GfaGraph.cpp:165-175: ids in order of first appearance
            if nm not in names:
                names[nm] = len(names)
            return names[nm]

        node_order = StdUnorderedOrder()      # GfaGraph::nodes  (unordered_map<int, string>, identity hash)
        edge_order = StdUnorderedOrder(lambda k: k[0] ^ (1 if k[1] else 0))   # GfaGraph::edges, hash<NodePos> = id ^ end (src/GfaGraph.h:27-31)
        seqs, targets = {}, {}
        for line in gfa_text.split("\n"):
            f = line.split()
            if not f or f[0] not in ("S", "L"):
                continue
            if f[0] == "S":
                i = name_id(f[1])
                node_order.insert(i)
                seqs[i] = f[2]
            else:
                a, b = name_id(f[1]), name_id(f[3])
                frm, to = (a, f[2] == "+"), (b, f[4] == "+")
                edge_order.insert(frm)
                targets.setdefault(frm, []).append(to)
        self.names = {v: k for k, v in names.items()}
        self.length, self.offset, self.ids, self.reverse, self.seq = [], [], [], [], []
        self.out, self.inn = [], []
        self.lookup = {}                      # bigraph id -> split nodes in offset order
        self.lookup_order = []
        ambiguous = []
        for i in node_order.order():
            for bid, s in ((2 * i, seqs[i]), (2 * i + 1, revcomp(seqs[i]))):   # ConvertGFANodeToNodes, src/BigraphToDigraph.cpp:101-104
                self.lookup[bid] = []
                for off in range(0, len(s), 64):                              # AddNode, src/AlignmentGraph.cpp:51-85
                    piece = s[off:off + 64]
                    self.lookup[bid].append(len(self.length))
                    self.length.append(len(piece)); self.offset.append(off); self.ids.append(bid); self.reverse.append(bid & 1); self.seq.append(piece.upper())
                    self.out.append([]); self.inn.append([])
                    ambiguous.append(any(c.upper() not in "ACGT" for c in piece))
                    if off > 0:
                        self.out[-2].append(len(self.out) - 1)
                        self.inn[-1].append(len(self.inn) - 2)
        for frm in edge_order.order():
            if frm[0] not in seqs:
                continue                      # edges of missing nodes are dropped (src/GfaGraph.cpp:296-330)
            for to in targets[frm]:
                if to[0] not in seqs:
                    continue
                # ConvertGFAEdgeToEdges, src/BigraphToDigraph.cpp:106-132
                from_left, from_right = (frm[0] * 2 + 1, frm[0] * 2) if frm[1] else (frm[0] * 2, frm[0] * 2 + 1)
                to_left, to_right = (to[0] * 2 + 1, to[0] * 2) if to[1] else (to[0] * 2, to[0] * 2 + 1)
                for a, b in ((from_right, to_right), (to_left, from_left)):
                    u, v = self.lookup[a][-1], self.lookup[b][0]               # AddEdgeNodeId, :233-253 (overlap 0)
                    if u not in self.inn[v]:
                        self.inn[v].append(u)
                    if v not in self.out[u]:
                        self.out[u].append(v)
        # RenumberAmbiguousToEnd, :919-974: non-ambiguous nodes keep their relative order, ambiguous ones go to the end in reverse
        n = len(self.length)
        ren, na, amb = [], 0, 0
        for i in range(n):
            if not ambiguous[i]:
                ren.append(na); na += 1
            else:
                ren.append(n - 1 - amb); amb += 1
        self.first_ambiguous = na

        def reorder(v):
            out = [None] * n
            for i, x in enumerate(v):
                out[ren[i]] = x
            return out
        self.length, self.offset, self.ids, self.reverse, self.seq = map(reorder, (self.length, self.offset, self.ids, self.reverse, self.seq))
        self.out = reorder([[ren[x] for x in l] for l in self.out])
        self.inn = reorder([[ren[x] for x in l] for l in self.inn])
        self.lookup = {k: [ren[x] for x in v] for k, v in self.lookup.items()}
        self.n = n

    def component_number(self):
        """doComponentOrder (Tarjan, :1008-1115) on a DAG: every node is its own component, numbered in DFS post-order from node 0
        upwards with out-neighbours in list order; componentNumber = N - 1 - post-order index."""
        post, seen, count = [0] * self.n, [False] * self.n, 0
        for root in range(self.n):
            if seen[root]:
                continue
            seen[root] = True
            stack = [(root, 0)]
            while stack:
                v, i = stack.pop()
                if i < len(self.out[v]):
                    stack.append((v, i + 1))
                    w = self.out[v][i]
                    if not seen[w]:
                        seen[w] = True
                        stack.append((w, 0))
                else:
                    post[v] = count
                    count += 1
        return [self.n - 1 - p for p in post]

    def weak_components(self):
        """buildComponentsMap, :1430-1463: BFS over out- then in-neighbours from the lowest unvisited node."""
        comp, idx, members = [None] * self.n, [None] * self.n, []
        for s in range(self.n):
            if comp[s] is not None:
                continue
            c, q = len(members), [s]
            comp[s], idx[s] = c, 0
            i = 0
            while i < len(q):
                v = q[i]; i += 1
                for t in self.out[v] + self.inn[v]:
                    if comp[t] is None:
                        comp[t], idx[t] = c, len(q)
                        q.append(t)
            members.append(q)
        return comp, idx, members

    def ancestors(self, v):
        """Nodes that reach v (v included)."""
        seen, q = {v}, deque([v])
        while q:
            x = q.popleft()
            for u in self.inn[x]:
                if u not in seen:
                    seen.add(u); q.append(u)
        return seen


# ---------------------------------------------------------------------------------------------- chaining, brute force

def chain_bruteforce(out_adj, in_adj, component_of, anchors):
    """Max-coverage chain as src/AlignmentGraph.cpp:1712-1863 defines it, by a quadratic DP over plain reachability.
    anchors: list of (path nodes, x, y). Anchor i may precede j when the last node of i strictly reaches the first node of j, or
    the two are the same node and i sorts before j by (y, x) (:1785-1822); and on the read either y_i < x_j (j adds its whole
    length, :1810,1839) or x_j <= y_i < y_j (j adds y_j - y_i, :1812,1841). C[j] = (covered bases, predecessor) maximised
    lexicographically (ties go to the larger anchor index), best end = max (C[j].first, j) (:1847-1849); components are tried in
    increasing id and a later one wins only with a strictly larger score (:1722-1733). Returns (chain, score)."""
    n = len(out_adj)
    reach_cache = {}

    def ancestors(v):
        if v not in reach_cache:
            seen, q = {v}, deque([v])
            while q:
                x = q.popleft()
                for u in in_adj[x]:
                    if u not in seen:
                        seen.add(u); q.append(u)
            reach_cache[v] = seen
        return reach_cache[v]

    # any topological order of the nodes
    indeg = [len(in_adj[v]) for v in range(n)]
    order, q = {}, deque(v for v in range(n) if indeg[v] == 0)
    while q:
        v = q.popleft()
        order[v] = len(order)
        for w in out_adj[v]:
            indeg[w] -= 1
            if indeg[w] == 0:
                q.append(w)
    by_comp = {}
    for j, (path, x, y) in enumerate(anchors):
        by_comp.setdefault(component_of[path[-1]], []).append(j)
    best_chain, best_score, first = [], 0, True
    for cid in sorted(by_comp):
        aids = by_comp[cid]
        C = {}
        todo = sorted(aids, key=lambda j: (order[anchors[j][0][0]], anchors[j][2], anchors[j][1]))
        rank = {j: r for r, j in enumerate(todo)}
        for j in todo:
            path_j, xj, yj = anchors[j]
            s_j = path_j[0]
            best = (yj - xj + 1, -1)
            anc = ancestors(s_j)
            for i in aids:
                if i == j or i not in C:
                    continue
                path_i, xi, yi = anchors[i]
                e_i = path_i[-1]
                if e_i == s_j:
                    if not ((yi, xi) < (yj, xj)):
                        continue
                elif e_i not in anc:
                    continue
                if yi < xj:
                    cand = (yj - xj + 1 + C[i][0], i)
                elif xj <= yi <= yj - 1:
                    cand = (yj + C[i][0] - yi, i)
                else:
                    continue
                best = max(best, cand)
            C[j] = best
        end = (0, -1)
        for j in aids:
            end = max(end, (C[j][0], j))
        chain, i = [], end[1]
        while i != -1:
            chain.append(i)
            i = C[i][1]
        chain.reverse()
        if first or end[0] > best_score:
            first, best_chain, best_score = False, chain, end[0]
    return best_chain, best_score


def find_linearizable(inn):
    """AlignmentGraph::findLinearizable, src/AlignmentGraph.cpp:644-735, statement for statement. The start node is marked `checked` before the walk begins (:662), so the walk's
    own "already checked" exit (:684-697) fires on the first iteration and clears the flag of the only node on the stack: the result is all False on every graph."""
    n = len(inn)
    linearizable, checked, on_stack = [False] * n, [False] * n, [False] * n
    for node in range(n):
        if checked[node]:
            continue
        checked[node] = True
        if len(inn[node]) != 1:
            continue
        stack = [node]
        on_stack[node] = True
        while True:
            back = stack[-1]
            if len(inn[back]) != 1 or checked[back]:
                for v in stack[:-1]:
                    checked[v], linearizable[v], on_stack[v] = True, True, False
                linearizable[back], checked[back], on_stack[back] = False, True, False
                break
            neighbor = inn[back][0]
            if neighbor == node:
                for v in stack:
                    checked[v], linearizable[v], on_stack[v] = True, False, False
                break
            if on_stack[neighbor]:
                i = len(stack) - 1
                while i > 0 and stack[i] != neighbor:
                    checked[stack[i]], linearizable[stack[i]], on_stack[stack[i]] = True, False, False
                    i -= 1
                for v in stack[:i]:
                    checked[v], linearizable[v], on_stack[v] = True, True, False
                break
            stack.append(neighbor)
            on_stack[neighbor] = True
    return linearizable
