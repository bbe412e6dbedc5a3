"""Independent model of ONE banded seed extension of the reference, in plain cell scores.

Test infrastructure. Written from the reference's sources (cited per function), NOT from oracle/: where the reference (and the oracle, and
the HIP kernels) keep a DP column as a Myers bit-vector triple (VP, VN, scoreEnd), this model keeps the 65 cell scores themselves
(row -1 = the last row of the slice above, rows 0..63 of the 64-row slice) and the plain edit-distance recurrence. What the bit-vector
operations MEAN in cells is pinned separately against the reference's own compiled WordSlice.h (tests/test_oracle_units.py: merge = cell-wise
minimum, getNextSlice = the recurrence, changedMinScore = minimum over the cells that got smaller). So a misreading of the band rule, the
node scheduling, the early exits, the top-row repair, the stop / trim logic or the backtrace rules shared by oracle/ and the kernels shows
up as a difference here.

Scope = the reference's configuration in chaining mode (src/AlignerMain.cpp:149,186-193): component priority queue, no ramp bandwidth,
unlimited cells per slice (scoresNotValid never set), approximate clipping, no forced global alignment, no X-drop.
"""
import math

import numpy as np

W = 64                      # rows per slice = WordConfiguration<uint64_t>::WordSize = AlignmentGraph::SPLIT_NODE_SIZE
INT_MAX = 2 ** 31 - 1

_IUPAC = {"A": "A", "C": "C", "G": "G", "T": "T", "U": "T", "N": "ACGT", "R": "AG", "Y": "CT", "K": "GT", "M": "CA", "S": "CG", "W": "AT",
          "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG"}


def character_match(sequence_char, graph_char):
    """src/GraphAlignerCommon.h:190-297 (characterMatch / ambiguousMatch): equal characters match; otherwise the two IUPAC sets must share a base; '-' matches nothing."""
    if sequence_char == graph_char:
        return True
    if sequence_char == "-" or graph_char == "-":
        return False
    return bool(set(_IUPAC[sequence_char.upper()]) & set(_IUPAC[graph_char.upper()]))


class Graph:
    """The split-node graph (A0) as plain lists; built from arrays the caller supplies (the graph build has its own model, tests/graph_model.py)."""

    def __init__(self, length, sequence, out_adj, in_adj, component, linearizable, node_ids, node_offset):
        self.length, self.sequence, self.out, self.inn = length, sequence, out_adj, in_adj
        self.component, self.linearizable, self.node_ids, self.node_offset = component, linearizable, node_ids, node_offset
        self.by_id = {}
        for v, big in enumerate(node_ids):
            self.by_id.setdefault(big, []).append(v)

    def unitig_node(self, bigraph_id, offset):
        """AlignmentGraph::GetUnitigNode: the split node of `bigraph_id` that holds `offset`."""
        for v in self.by_id[bigraph_id]:
            if self.node_offset[v] <= offset < self.node_offset[v] + self.length[v]:
                return v
        raise KeyError((bigraph_id, offset))


# ---- columns: numpy int64[65], index 0 = row -1, index r + 1 = row r ------------------------------------------------------------------

def source_column(score):
    """getSourceSliceFromScore, src/GraphAlignerBitvectorCommon.h:806-810: VP all ones over `score` = a column rising by one per row."""
    return np.arange(score, score + W + 1, dtype=np.int64)


def column_step(col, match, hin):
    """getNextSlice, ...Common.h:243-263 = one column of the edit-distance recurrence: the cell above this column's first row moves by `hin`,
    every cell is the minimum of left + 1, up + 1 and the diagonal + (0 on a match, else 1)."""
    new = np.empty(W + 1, dtype=np.int64)
    new[0] = col[0] + hin
    t = np.minimum(col[:-1] + np.where(match, 0, 1), col[1:] + 1)          # diagonal and horizontal candidates of rows 0..63
    rows = np.arange(W)
    run = np.minimum.accumulate(t - rows)                                  # vertical: new[r] = min over k <= r of t[k] + (r - k), and of top + r + 1
    new[1:] = rows + np.minimum(run, new[0] + 1)
    return new


def delta_words(col):
    """The (VP, VN) words of a column: bit r = the step from row r - 1 to row r is +1 / -1 (src/WordSlice.h)."""
    d = np.diff(col)
    vp = vn = 0
    for r in range(W):
        if d[r] == 1:
            vp |= 1 << r
        elif d[r] == -1:
            vn |= 1 << r
        else:
            assert d[r] == 0
    return vp, vn


def changed_min(new, old):
    """WordSlice::changedMinScore (cell-by-cell form, src/WordSlice.h:292-301): the smallest new score among the cells that got smaller."""
    smaller = new < old
    return int(new[smaller].min()) if smaller.any() else INT_MAX


class Item:
    """NodeSliceMapItemStruct, src/NodeSlice.h:15-47: a node's tile in one slice - first and last column, the horizontal steps of its last row, its minimum."""

    def __init__(self):
        self.start = np.full(W + 1, INT_MAX, dtype=np.int64)
        self.end = np.full(W + 1, INT_MAX, dtype=np.int64)
        self.exists = False
        self.bottom = [0] * W                  # HP - HN per column (column 0 unused)
        self.min_score = INT_MAX
        self.columns = None                    # every column of the last computation (the reference recomputes them on demand, ...Common.h:828-852)

    def copy(self):
        c = Item()
        c.start, c.end, c.exists, c.bottom, c.min_score, c.columns = self.start.copy(), self.end.copy(), self.exists, list(self.bottom), self.min_score, self.columns
        return c


def absent_previous():
    """src/GraphAlignerBitvectorBanded.h:313-321: a node that was not in the previous slice's band - its last row rises by one per column."""
    p = Item()
    p.bottom = [1] * W
    p.exists = False
    return p


class Slice:
    """DPSlice, ...Common.h:138-214 (the fields this configuration uses)."""

    def __init__(self):
        self.items = {}                        # node -> Item, in band-entry order (dict order). The reference iterates a hash map here; see the note in flatten_last_slice
        self.min_score = INT_MAX
        self.min_node = self.min_offset = None
        self.j = None
        self.bandwidth = 0
        self.correct_log, self.false_log = math.log(0.8), math.log(0.2)      # AlignmentCorrectnessEstimationState(), src/AlignmentCorrectnessEstimation.cpp:70-76
        self.correct_from_correct = self.false_from_correct = False

    def currently_correct(self):
        return self.correct_log > self.false_log


def _log_odds(mean, stddev):
    """getCorrectLogOdds / getWrongLogOdds, src/AlignmentCorrectnessEstimation.cpp:20-68."""
    v = [-(i - mean * W) ** 2 / (2 * (stddev * W) ** 2) for i in range(W // 2 + 1)]
    total = 0.0
    for x in v:
        total += math.exp(x)
    add = math.log(1.0 / total)
    v = [x + add for x in v]
    v += [v[-1]] * (W - W // 2)
    return v


_CORRECT_ODDS = _log_odds(0.1875, 0.0955)
_WRONG_ODDS = _log_odds(0.5, 0.0291)
_F2C, _F2F, _C2F, _C2C = math.log(0.00001), math.log(1.0 - 0.00001), math.log(0.0000000001), math.log(1.0 - 0.0000000001)


def next_correctness(prev, new, mismatches):
    """AlignmentCorrectnessEstimationState::NextState, src/AlignmentCorrectnessEstimation.cpp:105-129."""
    assert mismatches >= 0
    new.correct_from_correct = prev.correct_log + _C2C >= prev.false_log + _F2C
    new.false_from_correct = prev.correct_log + _C2F >= prev.false_log + _F2F
    c = max(prev.correct_log + _C2C, prev.false_log + _F2C)
    f = max(prev.correct_log + _C2F, prev.false_log + _F2F)
    idx = mismatches if mismatches < len(_CORRECT_ODDS) else len(_CORRECT_ODDS) - 1
    new.correct_log, new.false_log = c + _CORRECT_ODDS[idx], f + _WRONG_ODDS[idx]


class ModelAssertion(Exception):
    """One of the reference's assertions (they throw in its release build, src/ThreadReadAssertion.h:27)."""


def _check(cond, what):
    if not cond:
        raise ModelAssertion(what)


class _ComponentQueue:
    """src/ComponentPriorityQueue.h: a std::priority_queue of (component, score, node) with std::greater, one entry per active node, and the node's
    incoming edges ("extras") collected until it is popped. The heap follows libstdc++'s push_heap / pop_heap move for move, so that equal
    (component, score) keys - two nodes of one strongly connected component - leave in the same order."""

    def __init__(self):
        self.heap, self.active, self.extras = [], set(), {}

    @staticmethod
    def _greater(a, b):
        return a[0] > b[0] or (a[0] == b[0] and a[1] > b[1])

    def _push_heap(self, hole, top, value):
        h = self.heap
        parent = (hole - 1) // 2
        while hole > top and self._greater(h[parent], value):
            h[hole] = h[parent]
            hole = parent
            parent = (hole - 1) // 2
        h[hole] = value

    def insert(self, component, score, edge):
        node = edge[0]
        if node not in self.active:
            self.heap.append(None)
            self._push_heap(len(self.heap) - 1, 0, (component, score, node))
            self.active.add(node)
        self.extras.setdefault(node, []).append(edge)

    def top(self):
        return self.heap[0][2]

    def pop(self):
        h = self.heap
        node = h[0][2]
        self.extras[node] = []
        self.active.discard(node)
        value = h.pop()
        n = len(h)
        if n == 0:
            return
        hole, child = 0, 0                         # std::__adjust_heap
        while child < (n - 1) // 2:
            child = 2 * (child + 1)
            if self._greater(h[child], h[child - 1]):
                child -= 1
            h[hole] = h[child]
            hole = child
        if n % 2 == 0 and child == (n - 2) // 2:
            child = 2 * (child + 1)
            h[hole] = h[child - 1]
            hole = child - 1
        self._push_heap(hole, 0, value)

    def __len__(self):
        return len(self.heap)


class ExtensionModel:
    def __init__(self, graph, bandwidth):
        self.g, self.bandwidth = graph, bandwidth
        self._eq = {}
        self.fired = {}                        # how often each rule applied (the test wants every one of them exercised)

    def _fire(self, rule):
        self.fired[rule] = self.fired.get(rule, 0) + 1

    # -- matching one graph column against the 64 read rows of a slice: getEqVector, ...Common.h:280-319 (rows past the read's end match nothing)
    def _match(self, sequence, j, node, pos):
        c = self.g.sequence[node][pos]
        key = (j, c)
        if key not in self._eq:
            self._eq[key] = np.array([j + r < len(sequence) and character_match(sequence[j + r], c) for r in range(W)], dtype=bool)
        return self._eq[key].copy()

    # -- getInitialSliceExactPosition, ...Common.h:1243-1279: the row above the first slice is |column - seed column| on the seed's split node
    def initial_slice(self, bigraph_id, offset):
        s = Slice()
        s.j, s.bandwidth, s.min_score = -W, 1, 0
        node = self.g.unitig_node(bigraph_id, offset)
        inside = offset - self.g.node_offset[node]
        _check(inside < self.g.length[node], "offsetInNode < NodeLength")
        s.min_node, s.min_offset = node, inside
        it = Item()
        it.start = np.full(W + 1, inside, dtype=np.int64)                       # {0, 0, offsetInNode}: a flat column at that score
        it.end = np.full(W + 1, self.g.length[node] - 1 - inside, dtype=np.int64)
        it.min_score, it.exists = 0, True
        for i in range(1, self.g.length[node]):
            it.bottom[i] = -1 if i <= inside else 1
        s.items[node] = it
        return s

    # -- calculateNodeInner, ...Common.h:885-1168 (PreciseClipping = false). `early_leave` = AllowEarlyLeave; `prev_in_band` = bandCheck
    def calculate_node(self, node, item, prev, incoming, sequence, j, prev_in_band, early_leave=True):
        g = self.g
        length = g.length[node]
        ws = None
        has_skipless = False
        for (_, _, inc, skip_first) in incoming:                                  # :903-964
            if skip_first:
                ws = inc if ws is None else np.minimum(ws, inc)
                continue
            has_skipless = True
            if prev.exists:
                before = inc[0]
                hin = -1 if prev.start[W] < before else (1 if prev.start[W] > before else 0)
            else:
                hin = 1
            new = column_step(inc, self._match(sequence, j, node, 0), hin)
            if not prev.exists or new[0] < prev.start[W]:
                if prev.exists:
                    self._fire("entry below the row above")
                new[0] = new[1] + 1                                                # VP &= ~1, VN |= 1: the cell above is one more than the first row
            ws = new if ws is None else np.minimum(ws, new)
        _check(ws is not None, "hasWs")
        result = [int(ws[W]), 0]                                                   # minScore (of last-row cells), its column
        if item.exists:                                                            # :977-1050
            if has_skipless and len(g.inn[node]) == 1 and prev_in_band(g.inn[node][0]):
                if ws[W] > item.start[W]:
                    if early_leave:
                        self._fire("revisit: worse last row, left early")
                        return result, False
                elif ws[W] < item.start[W]:
                    self._fire("revisit: better last row, replaces")
                    pass                                                           # taken as smaller everywhere
                else:
                    nvp, nvn = delta_words(ws)
                    ovp, ovn = delta_words(item.start)
                    new_bigger = (nvp & ~ovp) | (ovn & ~nvn)
                    old_bigger = (ovp & ~nvp) | (nvn & ~ovn)
                    if new_bigger > old_bigger:
                        self._fire("revisit: equal last row, new word bigger")
                    elif old_bigger > new_bigger:
                        if early_leave:
                            self._fire("revisit: equal last row, old word bigger, left early")
                            return result, False
                    elif new_bigger == 0 and old_bigger == 0:
                        if early_leave:
                            self._fire("revisit: identical column, left early")
                            return result, False
                    else:
                        self._fire("revisit: equal last row, merged")
                        test = np.minimum(ws, item.start)
                        if np.array_equal(test, item.start) and early_leave:
                            return result, False
                        ws = test
            else:
                test = np.minimum(ws, item.start)
                tvp, _ = delta_words(test)
                ovp, ovn = delta_words(item.start)
                if test[W] == item.start[W] and tvp == ovp and tvp == ovn and early_leave:      # :1044 compares VP with VN (as written there)
                    return result, False
                if early_leave:
                    self._fire("revisit: several ways in, merged")
                ws = test
        if prev.exists and ws[0] > prev.start[W]:                                  # :1052-1058
            self._fire("entry above the row above: source column merged")
            ws = np.minimum(ws, source_column(int(prev.start[W])))
        top = list(prev.bottom)                                                    # previousSlice is a by-value copy: the repair below stays local
        force_until = 0
        if prev.exists:                                                            # :1068-1104: where this tile now enters cheaper than the row above says, the row above is
            before, comparison = int(ws[0]), int(prev.start[W])                    # replaced by a +1 ramp until the two meet
            _check(before <= comparison, "scoreBefore <= scoreComparison")
            if before < comparison:
                for fix in range(1, W):
                    new_comparison = comparison + top[fix]
                    _check(before <= new_comparison, "scoreBefore <= newScoreComparison")
                    if before < new_comparison:
                        if early_leave:
                            self._fire("top-row repair")
                        top[fix] = 1
                        force_until = fix
                    if before == new_comparison:
                        top[fix] = 0
                    before += 1
                    comparison = new_comparison
                    if before >= comparison:
                        break
        else:
            force_until = length
        item.start = ws
        item.exists = True
        item.bottom = [0] * W
        columns = [ws]
        for pos in range(1, length):                                               # :1118-1161
            match = self._match(sequence, j, node, pos)
            if not prev.exists:
                match[0] = False                                                   # forceEq: no free entry from a row that was never computed
            new = column_step(ws, match, top[pos])
            if force_until >= pos:
                new[0] = new[1] + 1
            item.bottom[pos] = int(new[W] - ws[W])
            ws = new
            if ws[W] < result[0]:
                result = [int(ws[W]), pos]
            columns.append(ws)
        item.end = ws
        item.columns = columns
        return result, True

    # -- recalcNodeWordslice, ...Common.h:828-852: the columns of a finished tile, from its first column and the row above
    def tile_columns(self, node, item, prev, sequence, j):
        copy = item.copy()
        self.calculate_node(node, copy, prev.copy(), [(node, 0, item.start, True)], sequence, j, lambda v: False, early_leave=False)
        _check(np.array_equal(copy.columns[0], item.start) and np.array_equal(copy.columns[-1], item.end), "recalc reproduces start and end")
        return copy.columns

    # -- flattenLastSliceEnd, ...Common.h:1170-1229: in the read's last, partial slice the minimum is taken at the read's last row
    def flatten_last_slice(self, cur, prev, sequence, j):
        offset = len(sequence) - j
        _check(0 <= offset < W, "partial slice")
        best = (INT_MAX, None, None)
        # NOTE the reference walks a hash map here (parallel-hashmap, absent from this image) and keeps the first strict minimum: ties between cells are
        # decided by that iteration order. Band-entry order is what oracle/ and the kernels define; with no tie any order gives the same answer.
        for node, item in cur.items.items():
            old = prev.items[node] if node in prev.items else absent_previous()
            cols = self.tile_columns(node, item, old, sequence, j)
            for i, col in enumerate(cols):
                flat = int(col[offset])                                            # flattenWordSlice(.., offset).scoreEnd = the score at row offset - 1
                if flat < best[0]:
                    best = (flat, node, i)
        _check(best[1] is not None, "minScore found")
        return best

    # -- calculateSlice, src/GraphAlignerBitvectorBanded.h:205-426 (component priority queue branch)
    def calculate_slice(self, sequence, j, cur, prev, prev_quit_score, bandwidth, prev_min_score):
        g = self.g
        queue = _ComponentQueue()
        for node, it in prev.items.items():                                       # :235-277
            if j == 0:
                _check(it.min_score <= prev_quit_score, "initial node inside the band")
            else:
                _check(it.exists, "previous item exists")
                if it.min_score > prev_quit_score:
                    self._fire("start: node outside the previous band")
                    continue
                if g.linearizable[node]:
                    nb = g.inn[node][0]
                    if nb in prev.items and prev.items[nb].end[W] < prev_quit_score and prev.items[nb].min_score < prev_quit_score:
                        self._fire("start: left to its only predecessor")
                        continue                                                   # its only predecessor will push it
            queue.insert(g.component[node], it.min_score, (node, it.min_score - prev_min_score, source_column(int(it.start[W])), True))
        _check(len(queue) > 0, "queue not empty")
        slice_min = INT_MAX - bandwidth - 1
        best = (slice_min, None, None)
        while len(queue) > 0:                                                      # :281-406
            node = queue.top()
            if not queue.extras.get(node):
                queue.pop()
                continue
            if node not in cur.items:
                cur.items[node] = Item()
            item = cur.items[node]
            old_end = item.end.copy() if item.exists else np.full(W + 1, INT_MAX, dtype=np.int64)
            prev_item = prev.items[node].copy() if node in prev.items else absent_previous()
            calc, _ = self.calculate_node(node, item, prev_item, list(queue.extras[node]), sequence, j, lambda v: v in prev.items)
            queue.pop()
            _check(calc[0] <= prev_quit_score + bandwidth + W + W, "node minimum inside the reachable range")
            slice_min = min(slice_min, calc[0])
            item.min_score = min(item.min_score, calc[0])                          # setMinScoreIfSmaller
            new_end = item.end
            if not np.array_equal(new_end, old_end):
                end_min = changed_min(new_end, old_end)
                _check(end_min >= prev_min_score and end_min != INT_MAX, "changed minimum")
                if end_min > slice_min + bandwidth:
                    self._fire("band rule: change not passed on")
                if end_min <= slice_min + bandwidth:                               # the band rule: only improvements within `bandwidth` of the slice's best last-row score travel on
                    for nb in g.out[node]:
                        queue.insert(g.component[nb], end_min, (nb, end_min - prev_min_score, new_end, False))
            if calc[0] < best[0]:
                best = (calc[0], node, calc[1])
            _check(best[0] == slice_min, "result.minScore == currentMinScoreAtEndRow")
        _check(best[1] is not None, "minScoreNode set")
        if j + W > len(sequence):                                                  # :414-417
            best = self.flatten_last_slice(cur, prev, sequence, j)
        return best

    # -- getViterbiSlices, ...Banded.h:513-701 without the ramp branch (rampBandwidth 0), + removeWronglyAlignedEnd, ...Common.h:1231-1241
    def slices(self, sequence, bigraph_id, offset):
        num_slices = (len(sequence) + W - 1) // W
        last = self.initial_slice(bigraph_id, offset)
        table = [last]
        _check(last.currently_correct(), "initial slice correct")
        for _ in range(num_slices):
            new = Slice()
            new.j = last.j + W
            best = self.calculate_slice(sequence, new.j, new, last, last.min_score + last.bandwidth, self.bandwidth, last.min_score)
            new.min_score, new.min_node, new.min_offset = best
            _check(new.min_score >= last.min_score, "slice minimum never falls")
            next_correctness(last, new, new.min_score - last.min_score)
            new.bandwidth = self.bandwidth
            if not new.correct_from_correct:                                       # :589-607: the best explanation of this slice no longer comes from "correct": stop here
                self._fire("stop: not correct-from-correct")
                break
            table.append(new)
            last = new
        currently_correct = table[-1].currently_correct()                          # removeWronglyAlignedEnd: drop slices back to where "wrong" branched off "correct"
        while not currently_correct:
            self._fire("trim: slice dropped")
            currently_correct = table[-1].false_from_correct
            table.pop()
            if not table:
                break
        return table

    # -- the backtrace: getReverseTraceFromTableStartLastRow / getReverseTraceFromTable, ...Common.h:385-544
    def trace(self, sequence, table):
        g = self.g
        last = table[-1]
        pos = (last.min_node, last.min_offset, min(last.j + W - 1, len(sequence) - 1))
        trace = [pos]
        current = (None, None)
        columns = None
        while trace[-1][2] != -1:
            node, offset, seq_pos = trace[-1]
            si = seq_pos // W + 1
            _check(si < len(table), "trace inside the table")
            cur, prev = table[si], table[si - 1]
            if current != (si, node):
                current = (si, node)
                _check(node in cur.items, "trace node in slice")
                columns = self.tile_columns(node, cur.items[node], prev.items[node] if node in prev.items else absent_previous(), sequence, cur.j)
            _check(offset < g.length[node], "offset inside node")
            if seq_pos % W == 0 and offset == 0:
                trace.append(self._corner(cur, prev, node, sequence))
                self._no_cycle(trace)
                continue
            if seq_pos % W == 0:
                if node not in prev.items:
                    self._fire("trace: first row of a node new in this slice")
                    trace.append((node, 0, seq_pos))
                    continue
                first, second = self._vertical_crossing(cur, prev, columns, node, trace[-1], sequence)
                if first[1] != trace[-1][1]:
                    for off in range(trace[-1][1] - 1, first[1], -1):
                        trace.append((first[0], off, first[2]))
                if first != trace[-1]:
                    trace.append(first)
                _check(second != trace[-1], "crossing moves")
                trace.append(second)
                continue
            if offset == 0:
                first, second = self._horizontal_crossing(cur, prev, node, trace[-1], sequence)
                if first[2] != trace[-1][2]:
                    for sp in range(trace[-1][2] - 1, first[2], -1):
                        trace.append((first[0], first[1], sp))
                if first != trace[-1]:
                    trace.append(first)
                _check(second != trace[-1], "crossing moves")
                trace.append(second)
                self._no_cycle(trace)
                continue
            trace.extend(self._inside(cur.j, columns, trace[-1], sequence))
        # the row above the first slice (:508-542): walk left while the seed ramp falls, then at most one step into an in-neighbour
        node = trace[-1][0]
        _check(node in table[0].items, "trace ends on an initial node")
        it = table[0].items[node]
        before = [int(it.start[W])]
        for i in range(1, g.length[node]):
            before.append(before[-1] + it.bottom[i])
        _check(before[-1] == int(it.end[W]), "ramp ends at endSlice")
        while before[trace[-1][1]] != 0 and trace[-1][1] > 0 and before[trace[-1][1] - 1] == before[trace[-1][1]] - 1:
            trace.append((node, trace[-1][1] - 1, trace[-1][2]))
        if trace[-1][1] == 0 and before[0] != 0:
            for nb in g.inn[node]:
                if nb in table[0].items and int(table[0].items[nb].end[0]) == before[0] - 1:
                    self._fire("trace: step into an in-neighbour above the first slice")
                    trace.append((nb, g.length[nb] - 1, trace[-1][2]))
                    break
        return last.min_score, trace

    @staticmethod
    def _no_cycle(trace):                                                          # checkBacktraceCircularity, :546-554
        for i in range(len(trace) - 2, -1, -1):
            _check(trace[i] != trace[-1], "trace does not revisit a cell")
            if trace[i][2] != trace[-1][2]:
                return

    def _inside(self, j, columns, pos, sequence):                                  # pickBacktraceInside, :556-597: up, then diagonal, then left
        node, hori, seq_pos = pos
        vert = seq_pos - j
        out = []
        while hori > 0 and vert > 0:
            here, up, left, diag = columns[hori][vert + 1], columns[hori][vert], columns[hori - 1][vert + 1], columns[hori - 1][vert]
            cost = 0 if character_match(sequence[vert + j], self.g.sequence[node][hori]) else 1
            _check(up >= here - 1 and left >= here - 1 and diag >= here - cost, "cell consistent with its neighbours")
            if up == here - 1:
                vert -= 1
            elif diag == here - cost:
                hori -= 1
                vert -= 1
            else:
                _check(left == here - 1, "some predecessor explains the cell")
                hori -= 1
            out.append((node, hori, vert + j))
        return out

    def _horizontal_crossing(self, cur, prev, node, pos, sequence):                # pickBacktraceHorizontalCrossing, :599-663 (first column of a node, not the first row)
        g = self.g
        start = cur.items[node].start
        n, o, sp = pos
        while sp % W != 0 and start[sp % W + 1] - start[sp % W] == 1:              # VP bit set: came from above
            sp -= 1
        off = sp % W
        if off == 0:
            return (n, o, sp), self._corner(cur, prev, node, sequence)
        cost = 0 if character_match(sequence[sp], g.sequence[n][o]) else 1
        here = int(start[off + 1])
        quit_score = cur.min_score + cur.bandwidth
        if here > quit_score:                                                      # outside the band: the smallest neighbour cell, whatever its score
            self._fire("trace: node crossing outside the band")
            smallest, where = int(start[off]), (node, 0, sp - 1)
            for nb in g.inn[node]:
                if nb not in cur.items:
                    continue
                end = cur.items[nb].end
                if end[off] <= smallest:
                    smallest, where = int(end[off]), (nb, g.length[nb] - 1, sp - 1)
                if end[off + 1] < smallest and nb != node:
                    smallest, where = int(end[off + 1]), (nb, g.length[nb] - 1, sp)
            _check(where != (n, o, sp), "crossing moves")
            return (n, o, sp), where
        for nb in g.inn[node]:
            if nb not in cur.items:
                continue
            end = cur.items[nb].end
            _check(end[off + 1] >= here - 1 and end[off] >= here - cost, "neighbour consistent")
            if end[off + 1] == here - 1:
                return (n, o, sp), (nb, g.length[nb] - 1, sp)
            if end[off] == here - cost:
                return (n, o, sp), (nb, g.length[nb] - 1, sp - 1)
        raise ModelAssertion("horizontal crossing: no predecessor")

    def _vertical_crossing(self, cur, prev, columns, node, pos, sequence):         # pickBacktraceVerticalCrossing, :665-708 (first row of a slice, not the first column)
        n, o, sp = pos
        while o > 0 and columns[o - 1][1] == columns[o][1] - 1:
            o -= 1
        if o == 0:
            return (n, o, sp), self._corner(cur, prev, node, sequence)
        _check(node in prev.items, "node in previous slice")
        cost = 0 if character_match(sequence[sp], self.g.sequence[n][o]) else 1
        p = prev.items[node]
        here = int(columns[o][1])
        diagonal = int(p.start[W]) + sum(p.bottom[1:o])
        up = diagonal + p.bottom[o]
        if here > cur.min_score + cur.bandwidth or diagonal > prev.min_score + prev.bandwidth or up > prev.min_score + prev.bandwidth:
            self._fire("trace: slice crossing outside the band")
            return (n, o, sp), ((n, o - 1, sp - 1) if diagonal < up else (n, o, sp - 1))
        _check(up >= here - 1 and diagonal >= here - cost, "row above consistent")
        if up == here - 1:
            return (n, o, sp), (n, o, sp - 1)
        _check(diagonal == here - cost, "diagonal explains the cell")
        return (n, o, sp), (n, o - 1, sp - 1)

    def _corner(self, cur, prev, node, sequence):                                  # pickBacktraceCorner, :710-804 (first row and first column)
        g = self.g
        j = cur.j
        here = int(cur.items[node].start[1])
        if here > cur.min_score + cur.bandwidth:
            self._fire("trace: corner outside the band")
            smallest, where = here + 1, (0, 0, 0)
            if node in prev.items:
                smallest, where = int(prev.items[node].start[W]), (node, 0, j - 1)
            for nb in g.inn[node]:
                if nb in prev.items and prev.items[nb].end[W] <= smallest:
                    smallest, where = int(prev.items[nb].end[W]), (nb, g.length[nb] - 1, j - 1)
                if nb in cur.items and nb != node and cur.items[nb].end[1] < smallest:
                    smallest, where = int(cur.items[nb].end[1]), (nb, g.length[nb] - 1, j)
            return where
        cost = 0 if character_match(sequence[j], g.sequence[node][0]) else 1
        if node in prev.items:
            _check(prev.items[node].start[W] >= here - 1, "cell above consistent")
            if prev.items[node].start[W] == here - 1:
                return (node, 0, j - 1)
        best_invalid, best_invalid_score = None, here + 1
        for nb in g.inn[node]:
            if nb in cur.items:
                _check(cur.items[nb].end[1] >= here - 1, "left neighbour consistent")
                if cur.items[nb].end[1] == here - 1:
                    return (nb, g.length[nb] - 1, j)
            if nb in prev.items:
                corner = int(prev.items[nb].end[W])
                if corner > prev.min_score + prev.bandwidth:
                    self._fire("trace: corner candidate outside the previous band")
                    if corner < best_invalid_score:
                        best_invalid, best_invalid_score = (nb, g.length[nb] - 1, j - 1), corner
                else:
                    _check(corner >= here - cost, "corner consistent")
                    if corner == here - cost:
                        return (nb, g.length[nb] - 1, j - 1)
        if best_invalid_score < here + 1:
            return best_invalid
        raise ModelAssertion("corner: no predecessor")

    # -- getReverseTraceFromSeed, ...Banded.h:46-71
    def extend(self, sequence, bigraph_id, offset):
        self._eq = {}
        table = self.slices(sequence, bigraph_id, offset)
        out = {
            "slice_min": [s.min_score for s in table],
            "slice_min_cell": [(s.min_node, s.min_offset) for s in table],
            "slice_nodes": [sorted(s.items) for s in table],
            "failed": len(table) <= 1, "score": None, "trace": [],
        }
        if len(table) > 1:
            out["score"], out["trace"] = self.trace(sequence, table)
        return out
