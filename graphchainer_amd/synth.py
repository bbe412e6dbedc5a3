"""Deterministic synthetic variation graphs and long reads.

The GPU box has no genome data, so BASELINE.json's configurations are instantiated synthetically:
graph = random ACGT backbone cut at variant sites (Poisson, mean spacing 45 bp; 85 % SNP bubbles with two
1-bp allele nodes, 15 % indel bubbles with one 1-20 bp node and a bypass edge) written as GFA S/L lines
with integer names in file order; reads = random haplotype walks, either strand, fixed length, i.i.d.
ONT-like errors (3 % deletion, 4 % substitution, 3 % 1-bp insertion). The read model follows the spirit
of the reference's src/SimulateReads.cpp:13-42 (uniform start, per-base error draws).
"""
import numpy as np

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGT", b"TGCA"):
    _COMP[_a] = _b


class SynthGraph:
    """Backbone + variant sites; can write itself as GFA and sample haplotype reads."""

    def __init__(self, backbone_len, seed=7, mean_spacing=45.0, snp_fraction=0.85, max_indel=20):
        rng = np.random.default_rng(seed)
        self.backbone = _BASES[rng.integers(0, 4, size=backbone_len)]
        # variant site positions: gaps ~ 1 + geometric with the requested mean
        n_est = int(backbone_len / mean_spacing * 1.2) + 16
        gaps = rng.geometric(1.0 / mean_spacing, size=n_est) + 1
        pos = np.cumsum(gaps)
        pos = pos[pos < backbone_len - 2]
        self.site_pos = pos.astype(np.int64)              # backbone index of the site (SNP base / indel insertion point)
        n = len(pos)
        self.is_snp = rng.random(n) < snp_fraction
        alt = (np.searchsorted(_BASES, self.backbone[pos]) + rng.integers(1, 4, size=n)) % 4
        self.snp_alt = _BASES[alt]
        indel_len = np.minimum(rng.geometric(0.25, size=n), max_indel)
        self.indel_len = np.where(self.is_snp, 0, indel_len).astype(np.int64)
        self.indel_off = np.concatenate([[0], np.cumsum(self.indel_len)])
        self.indel_seq = _BASES[rng.integers(0, 4, size=int(self.indel_off[-1]))]
        # backbone segment i = backbone[seg_start[i]:seg_end[i]]; an SNP site consumes its own backbone base
        consumed = self.is_snp.astype(np.int64)
        self.seg_start = np.concatenate([[0], pos + consumed])
        self.seg_end = np.concatenate([pos, [backbone_len]])
        self.backbone_len = backbone_len

    def write_gfa(self, path):
        """Segments are named 1..N in file order; links carry 0M overlaps."""
        bb = self.backbone.tobytes()
        ins = self.indel_seq.tobytes()
        out = []
        links = []
        next_id = 1
        prev = None
        n = len(self.site_pos)
        for i in range(n + 1):
            seg = bb[self.seg_start[i]:self.seg_end[i]]
            sid = next_id
            next_id += 1
            out.append(b"S\t%d\t%s\n" % (sid, seg))
            if prev is not None:
                for p in prev:
                    links.append(b"L\t%d\t+\t%d\t+\t0M\n" % (p, sid))
            if i == n:
                break
            if self.is_snp[i]:
                a, b = next_id, next_id + 1
                next_id += 2
                p = int(self.site_pos[i])
                out.append(b"S\t%d\t%s\n" % (a, bb[p:p + 1]))
                out.append(b"S\t%d\t%s\n" % (b, bytes([int(self.snp_alt[i])])))
                links.append(b"L\t%d\t+\t%d\t+\t0M\n" % (sid, a))
                links.append(b"L\t%d\t+\t%d\t+\t0M\n" % (sid, b))
                prev = (a, b)
            else:
                a = next_id
                next_id += 1
                out.append(b"S\t%d\t%s\n" % (a, ins[self.indel_off[i]:self.indel_off[i + 1]]))
                links.append(b"L\t%d\t+\t%d\t+\t0M\n" % (sid, a))
                prev = (sid, a)
        with open(path, "wb") as f:
            f.writelines(out)
            f.writelines(links)
        return next_id - 1

    def haplotype_window(self, rng, start, length):
        """Sequence of a random haplotype starting at backbone coordinate `start`, at least `length` long."""
        i = int(np.searchsorted(self.seg_end, start, side="right"))
        pieces = []
        total = 0
        at = max(start, int(self.seg_start[i])) if i < len(self.seg_start) else start
        n = len(self.site_pos)
        while total < length and i <= n:
            seg = self.backbone[at:self.seg_end[i]]
            pieces.append(seg)
            total += len(seg)
            if i == n:
                break
            if self.is_snp[i]:
                if rng.random() < 0.5:
                    pieces.append(self.backbone[self.site_pos[i]:self.site_pos[i] + 1])
                else:
                    pieces.append(self.snp_alt[i:i + 1])
                total += 1
            elif rng.random() < 0.5:
                pieces.append(self.indel_seq[self.indel_off[i]:self.indel_off[i + 1]])
                total += int(self.indel_len[i])
            i += 1
            at = int(self.seg_start[i])
        return np.concatenate(pieces)[:length] if pieces else np.zeros(0, dtype=np.uint8)

    def sample_reads(self, n_reads, read_len, seed=11, p_del=0.03, p_sub=0.04, p_ins=0.03):
        """Returns a list of bytes. ONT-like defaults; CLR-like is (0.04, 0.02, 0.09)."""
        rng = np.random.default_rng(seed)
        reads = []
        span = int(read_len * 1.15) + 64
        for _ in range(n_reads):
            start = int(rng.integers(0, max(1, self.backbone_len - span)))
            hap = self.haplotype_window(rng, start, span)
            u = rng.random(len(hap))
            keep = u >= p_del
            sub = (u >= p_del) & (u < p_del + p_sub)
            seq = hap.copy()
            if sub.any():
                codes = (np.searchsorted(_BASES, seq[sub]) + rng.integers(1, 4, size=int(sub.sum()))) % 4
                seq[sub] = _BASES[codes]
            ins = (u >= p_del + p_sub) & (u < p_del + p_sub + p_ins)
            seq = seq[keep]
            ins = ins[keep]
            if ins.any():
                where = np.nonzero(ins)[0] + 1
                seq = np.insert(seq, where, _BASES[rng.integers(0, 4, size=len(where))])
            seq = seq[:read_len]
            if rng.random() < 0.5:
                seq = _COMP[seq[::-1]]
            reads.append(seq.tobytes())
        return reads


def write_fasta(path, reads):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b">read%d\n%s\n" % (i, r))
