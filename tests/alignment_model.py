"""A second reading of AlignOneWay, the two-directional trace and its fix-ups, and the anchor construction, written from the
reference's sources in plain Python; nothing is imported from oracle/ or the product. The DP of one extension is tests/extension_model.py (rows A5-A10).

  exact_alignment_part       GraphAligner::exactAlignmentPart, src/GraphAligner.h:407-461 (the interpolated binary search as written, then the two scans)
  two_directional_trace      getTwoDirectionalTrace, :486-525
  fix_forward / fix_reverse  fixForwardTraceSeqPos :527-540, fixReverseTraceSeqPosAndOrder :543-565
  alignment_from_seed        getAlignmentFromSeed, :567-626
  align_one_way              AlignOneWay, :114-203 (seedExtendDensity = -1, nondeterministicOptimizations off: src/AlignerMain.cpp:186-193)
  anchors_of_read            the fragment loop and anchor construction of runComponentMappings, src/Aligner.cpp:667-729

A trace item is (bigraph node id, offset in the original node, read position, nodeSwitch). On a DAG a backtrace step carries nodeSwitch exactly when it enters another split node
(pickBacktrace*: the second member of the returned pair is set by the branches that step into an in-neighbour, src/GraphAlignerBitvectorCommon.h:599-804), which is how
the flag is derived from the extension model's cell list here."""
_COMPLEMENT = {ord(a): ord(b) for a, b in zip("ACGTUacgtuRYKMSWBDHVNrykmswbdhvn", "TGCAAtgcaaYRMKSWVHDBNyrmkswvhdbn")}


def reverse_complement(seq):
    return bytes(_COMPLEMENT[c] for c in reversed(seq))


class AlignmentModel:
    def __init__(self, extension_model, graph, original_size):
        """extension_model: ExtensionModel over `graph` (extension_model.Graph); original_size: {bigraph node id: length of the original node}."""
        self.ext, self.g, self.original_size = extension_model, graph, original_size

    def reverse_position(self, node_id, offset):            # AlignmentGraph::GetReversePosition, src/AlignmentGraph.cpp:850-868
        return node_id ^ 1, self.original_size[node_id] - offset - 1

    # ---- one extension, as getReverseTraceFromSeed returns it: cells from the alignment's far end to the row above the first slice, with their nodeSwitch flags
    def _one_way(self, sequence, bigraph_id, offset):
        r = self.ext.extend(sequence.decode(), bigraph_id, offset)
        if r["failed"]:
            return None
        cells = r["trace"]
        items = [(node, off, sp, i > 0 and cells[i - 1][0] != node) for i, (node, off, sp) in enumerate(cells)]
        return r["score"], items

    def two_directional_trace(self, sequence, rev_sequence, seed):
        forward_id = seed["nodeID"] * 2 + (1 if seed["reverse"] else 0)
        backward_id = forward_id ^ 1
        p = seed["seqPos"]
        backward = forward = None
        if p > 0:
            rev = self.reverse_position(forward_id, seed["nodeOffset"])
            assert rev[0] == backward_id
            backward = self._one_way(rev_sequence[len(rev_sequence) - p:], backward_id, rev[1])
        if p < len(sequence) - 1:
            forward = self._one_way(sequence[p + 1:], forward_id, seed["nodeOffset"])
        # both traces are reversed: they now run from the row above the first slice outwards
        if backward:
            backward = (backward[0], backward[1][::-1])
        if forward:
            forward = (forward[0], forward[1][::-1])
        return backward, forward

    def fix_forward(self, items, start):
        return [(self.g.node_ids[node], off + self.g.node_offset[node], (sp + start), sw) for node, off, sp, sw in items]

    def fix_reverse(self, items, end):
        items = items[::-1]
        out = []
        for node, off, sp, sw in items:
            rnode, roff = self.reverse_position(self.g.node_ids[node], self.g.node_offset[node] + off)
            out.append([rnode, roff, end - sp, sw])             # (size_t arithmetic: the row above the first slice, -1, lands on end + 1 = the seed's position)
        for i in range(len(out) - 1):
            out[i][3] = out[i + 1][3]
        out[-1][3] = False
        return [tuple(x) for x in out]

    def alignment_from_seed(self, sequence, rev_sequence, seed):
        """None when both directions fail (alignmentFailed), else {trace, score, start, end}."""
        backward, forward = self.two_directional_trace(sequence, rev_sequence, seed)
        btrace = self.fix_reverse(backward[1], seed["seqPos"] - 1) if backward else []
        ftrace = self.fix_forward(forward[1], seed["seqPos"] + 1) if forward else []
        if not backward and not forward:
            return None
        if not backward:
            trace, score = ftrace, forward[0]
        elif forward:
            assert btrace[-1][:3] == ftrace[0][:3], "the two halves meet on the seed cell"
            trace, score = btrace[:-1] + ftrace, backward[0] + forward[0]
        else:
            trace, score = btrace, backward[0]
        return {"trace": trace, "score": score, "start": trace[0][2], "end": trace[-1][2] + 1}

    @staticmethod
    def exact_alignment_part(aln, seed):
        trace = aln["trace"]
        sp = seed["seqPos"]
        if trace[-1][2] < sp or trace[0][2] > sp:
            return False
        high, low = len(trace), 0
        mid = (sp - trace[0][2]) // (trace[-1][2] - trace[0][2])
        while trace[mid][2] != sp:
            if trace[mid][2] < sp:
                low = mid
                mid = (high + low) // 2
                if mid == low:
                    mid += 1
                assert mid < len(trace)
            if trace[mid][2] > sp:
                high = mid
                mid = (high + low) // 2
            assert low < mid < high
        compare_node = seed["nodeID"] * 2 + (1 if seed["reverse"] else 0)
        down = mid
        while trace[down][2] == sp:
            if trace[down][0] == compare_node and trace[down][1] == seed["nodeOffset"]:
                return True
            if down == 0:
                break
            down -= 1
        up = mid
        while trace[up][2] == sp:
            if trace[up][0] == compare_node and trace[up][1] == seed["nodeOffset"]:
                return True
            up += 1
            if up == len(trace):
                break
        return False

    def align_one_way(self, sequence, seeds, sloppy, min_cluster_size=1, l=0, r=None, offset=0):
        """AlignOneWay over seeds[l:r] (all of them in the whole-read call); `sequence` is the read or the fragment, seed positions are shifted by `offset`.
        Returns (alignments in the list's final order, seeds extended)."""
        r = len(seeds) if r is None else r
        rev_sequence = reverse_complement(sequence)
        alignments, extended = [], 0
        end_to_end_score = 0
        extend_seeds = len(seeds)                               # seedExtendDensity == -1
        worst_extended = 0
        for i in range(l, min(len(seeds), r)):
            if sloppy and seeds[i]["goodness"] < end_to_end_score:
                break
            if extended >= extend_seeds and seeds[i]["goodness"] < worst_extended:
                break
            seed = dict(seeds[i])
            seed["seqPos"] -= offset
            if seed["cluster"] < min_cluster_size:
                continue
            if sloppy and any(a["start"] <= seed["seqPos"] <= a["end"] and a["goodness"] > seed["goodness"] for a in alignments):
                continue
            if any(self.exact_alignment_part(a, seed) for a in alignments):
                continue
            worst_extended = seed["goodness"]
            extended += 1
            item = self.alignment_from_seed(sequence, rev_sequence, seed)
            if item is None:
                continue
            item["goodness"] = seed["goodness"]
            alignments.append(item)
            if sloppy:
                alignments.sort(key=lambda a: a["start"])       # (std::sort by alignmentStart: what follows reads the list in that order, ties do not change its outcome)
                if alignments[0]["start"] == 0:
                    min_goodness, contiguous_end = alignments[0]["goodness"], alignments[0]["end"]
                    for a in alignments[1:]:
                        if a["start"] <= contiguous_end:
                            min_goodness = min(min_goodness, a["goodness"])
                            contiguous_end = max(contiguous_end, a["end"])
                    if contiguous_end == len(sequence):
                        end_to_end_score = min_goodness
        return alignments, extended

    def anchors_of_read(self, sequence, seeds_by_position, split_len=35, split_gap=35):
        """The fragment loop of runComponentMappings: every fragment's alignments (non-sloppy AlignOneWay over the window's seeds) become anchors.
        Returns [(x, y, path of split nodes, first cell, last cell, score)] with the cells as (split node, offset in it, read position)."""
        anchors = []
        sl = sr = 0
        for l in range(0, len(sequence) - split_len + 1, split_gap):
            while sr < len(seeds_by_position) and seeds_by_position[sr]["seqPos"] + seeds_by_position[sr]["matchLen"] <= l + split_len:
                sr += 1
            while sl < sr and seeds_by_position[sl]["seqPos"] < l:
                sl += 1
            if sl >= sr:
                continue
            alignments, _ = self.align_one_way(sequence[l:l + split_len], seeds_by_position, False, l=sl, r=sr, offset=l)
            for a in alignments:
                path = []
                for node, off, _sp, _sw in a["trace"]:
                    split = self.g.unitig_node(node, off)
                    if not path or split != path[-1]:
                        path.append(split)
                ends = []
                for node, off, sp, _sw in (a["trace"][0], a["trace"][-1]):
                    split = self.g.unitig_node(node, off)
                    ends.append((split, off - self.g.node_offset[split], sp + l))
                anchors.append((l, l + split_len - 1, path, ends[0], ends[1], a["score"]))
        return anchors
