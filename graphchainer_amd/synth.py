"""Deterministic synthetic variation graphs and long reads.

The GPU box has no genome data, so BASELINE.json's configurations are instantiated synthetically:
graph = random ACGT backbone cut at variant sites (Poisson, mean spacing 45 bp; 85 % SNP bubbles with two
1-bp allele nodes, 15 % indel bubbles with one 1-20 bp node and a bypass edge) written as GFA S/L lines
with integer names in file order; reads = random haplotype walks, either strand, fixed length, i.i.d.
ONT-like errors (3 % deletion, 4 % substitution, 3 % 1-bp insertion). The read model follows the spirit
of the reference's src/SimulateReads.cpp:13-42 (uniform start, per-base error draws).

Optional structure for the configurations a plain SNP/indel chain does not reach (all off by default, so the default graphs
and reads are unchanged): multi-allelic SNP sites (3-4 parallel allele nodes: path-cover width > 2), nested bubbles (an
insertion that itself carries a SNP), links written from the reverse strand ("L b - a -"), repeats (copies of a few
windows pasted elsewhere in the backbone with a little divergence: many seeds per fragment), and SynthGenome: several
such chromosomes in one GFA (separate weakly connected components, BASELINE config 5's shape).
"""
import os
import shutil

import numpy as np

_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGT", b"TGCA"):
    _COMP[_a] = _b


class SynthGraph:
    """Backbone + variant sites; can write itself as GFA and sample haplotype reads."""

    def __init__(self, backbone_len, seed=7, mean_spacing=45.0, snp_fraction=0.85, max_indel=20,
                 multi_allelic=0.0, nested=0.0, repeats=0, repeat_len=2000, repeat_divergence=0.02, minus_links=0.0):
        rng = np.random.default_rng(seed)
        self.backbone = _BASES[rng.integers(0, 4, size=backbone_len)]
        self.minus_links = minus_links
        self.seed = seed
        if repeats:
            # paste diverged copies of a few source windows elsewhere (own generator: the default graph's draws are untouched)
            rrng = np.random.default_rng(seed + 1000003)
            repeat_len = min(repeat_len, backbone_len // 4)
            n_sources = max(1, repeats // 3)
            sources = rrng.integers(0, backbone_len - repeat_len, size=n_sources)
            for c in range(repeats):
                src = int(sources[c % n_sources])
                dst = int(rrng.integers(0, backbone_len - repeat_len))
                copy = self.backbone[src:src + repeat_len].copy()
                hit = rrng.random(repeat_len) < repeat_divergence
                copy[hit] = _BASES[rrng.integers(0, 4, size=int(hit.sum()))]
                self.backbone[dst:dst + repeat_len] = copy
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
        # optional site structure, from a generator of its own
        xrng = np.random.default_rng(seed + 2000003)
        extra = np.where(self.is_snp & (xrng.random(n) < multi_allelic), xrng.integers(1, 3, size=n), 0)   # 1-2 further alleles
        self.n_alleles = (2 + extra).astype(np.int64)                                                          # SNP sites: 2..4 allele nodes
        ref_code = np.searchsorted(_BASES, self.backbone[pos])
        shift = (alt - ref_code) % 4                                                                           # 1..3: the first alt allele
        self.snp_alts = np.stack([_BASES[(ref_code + ((shift - 1 + k) % 3) + 1) % 4] for k in range(3)], axis=1)   # all three non-reference bases, first = snp_alt
        self.nested = (~self.is_snp) & (self.indel_len >= 3) & (xrng.random(n) < nested)                       # insertion with a SNP inside it
        self.nested_at = np.where(self.nested, 1 + xrng.integers(0, 1 << 30, size=n) % np.maximum(self.indel_len - 2, 1), 0)
        self.nested_alt = _BASES[xrng.integers(0, 4, size=n)]

    def gfa_lines(self, first_id=1):
        """(segment lines, link lines, next free id). Segments are named first_id.. in file order; links carry 0M overlaps."""
        bb = self.backbone.tobytes()
        ins = self.indel_seq.tobytes()
        out = []
        links = []
        next_id = first_id
        prev = None
        n = len(self.site_pos)
        lrng = np.random.default_rng(self.seed + 3000003)
        flip = lrng.random(8 * (n + 2)) < self.minus_links if self.minus_links > 0 else None
        n_links = 0

        def link(a, b):
            nonlocal n_links
            if flip is not None and flip[n_links % len(flip)]:
                links.append(b"L\t%d\t-\t%d\t-\t0M\n" % (b, a))     # the same adjacency written from the reverse strand
            else:
                links.append(b"L\t%d\t+\t%d\t+\t0M\n" % (a, b))
            n_links += 1

        def node(seq):
            nonlocal next_id
            out.append(b"S\t%d\t%s\n" % (next_id, seq))
            next_id += 1
            return next_id - 1

        for i in range(n + 1):
            sid = node(bb[self.seg_start[i]:self.seg_end[i]])
            if prev is not None:
                for p in prev:
                    link(p, sid)
            if i == n:
                break
            if self.is_snp[i]:
                p = int(self.site_pos[i])
                alleles = [node(bb[p:p + 1])] + [node(bytes([int(self.snp_alts[i, k])])) for k in range(int(self.n_alleles[i]) - 1)]
                for a in alleles:
                    link(sid, a)
                prev = tuple(alleles)
            elif self.nested[i]:
                seq = ins[self.indel_off[i]:self.indel_off[i + 1]]
                at = int(self.nested_at[i])
                pre, post = node(seq[:at]), None
                x, y = node(seq[at:at + 1]), node(bytes([int(self.nested_alt[i])]))
                post = node(seq[at + 1:])
                link(sid, pre); link(pre, x); link(pre, y); link(x, post); link(y, post)
                prev = (sid, post)
            else:
                a = node(ins[self.indel_off[i]:self.indel_off[i + 1]])
                link(sid, a)
                prev = (sid, a)
        return out, links, next_id

    def write_gfa(self, path):
        out, links, next_id = self.gfa_lines(1)
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
                elif self.n_alleles[i] == 2:
                    pieces.append(self.snp_alt[i:i + 1])
                else:
                    k = int(rng.integers(0, self.n_alleles[i] - 1))
                    pieces.append(self.snp_alts[i, k:k + 1])
                total += 1
            elif rng.random() < 0.5:
                seq = self.indel_seq[self.indel_off[i]:self.indel_off[i + 1]]
                if self.nested[i] and rng.random() < 0.5:
                    seq = seq.copy()
                    seq[int(self.nested_at[i])] = self.nested_alt[i]
                pieces.append(seq)
                total += int(self.indel_len[i])
            i += 1
            at = int(self.seg_start[i])
        return np.concatenate(pieces)[:length] if pieces else np.zeros(0, dtype=np.uint8)

    def sample_reads(self, n_reads, read_len, seed=11, p_del=0.03, p_sub=0.04, p_ins=0.03, sv_fraction=0.0, sv_len=1500):
        """Returns a list of bytes. ONT-like defaults; CLR-like is (0.04, 0.02, 0.09). sv_fraction of the reads carry a deletion of
        sv_len graph bases in their middle (a structural variant the graph does not hold): the whole-read aligner stops at the
        breakpoint while the chain bridges it - the reads GraphChainer's chained alignment is for."""
        rng = np.random.default_rng(seed)
        sv_rng = np.random.default_rng(seed + 4000003) if sv_fraction > 0 else None
        reads = []
        span = int(read_len * 1.15) + 64
        for _ in range(n_reads):
            with_sv = sv_rng is not None and sv_rng.random() < sv_fraction
            take = span + (sv_len if with_sv else 0)
            start = int(rng.integers(0, max(1, self.backbone_len - take)))
            hap = self.haplotype_window(rng, start, take)
            if with_sv and len(hap) > read_len // 2 + sv_len:
                hap = np.concatenate([hap[:read_len // 2], hap[read_len // 2 + sv_len:]])
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


class SynthGenome:
    """Several SynthGraph chromosomes in one GFA: separate weakly connected components (two per chromosome, one per strand)."""

    def __init__(self, n_chromosomes, backbone_len, seed=7, **kw):
        self.chromosomes = [SynthGraph(backbone_len, seed=seed + 7919 * c, **kw) for c in range(n_chromosomes)]
        self.backbone_len = backbone_len

    def write_gfa(self, path):
        """All S lines, then all L lines (the reference's node numbering follows the order of first appearance). One chromosome's lines in memory at a time: its
        links wait in a file of their own (at 3.1 Gbp the lines of the whole genome are ~45 GB of Python objects)."""
        first = 1
        links_path = path + ".links"
        with open(path, "wb") as f, open(links_path, "wb") as lf:
            for chrom in self.chromosomes:
                s, l, first = chrom.gfa_lines(first)
                f.writelines(s)
                lf.writelines(l)
                del s, l
        with open(path, "ab") as f, open(links_path, "rb") as lf:
            shutil.copyfileobj(lf, f, 64 << 20)
        os.remove(links_path)
        return first - 1

    def sample_reads(self, n_reads, read_len, seed=11, **kw):
        """Reads drawn round-robin from the chromosomes (each with its own stream)."""
        per = [c.sample_reads((n_reads - i + len(self.chromosomes) - 1) // len(self.chromosomes), read_len, seed=seed + 31 * i, **kw) for i, c in enumerate(self.chromosomes)]
        out = []
        for k in range(n_reads):
            out.append(per[k % len(per)][k // len(per)])
        return out


def write_fasta(path, reads):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b">read%d\n%s\n" % (i, r))
