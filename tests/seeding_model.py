"""A second reading of the seeding stages, written from the reference's sources in plain Python; nothing is imported from oracle/ or the product.

  iterate_kmers            iterateKmers, src/MinimizerSeeder.cpp:59-102 (the read's k-mers with the thinning rule, restarts after non-ACGT letters, size_t wrap-around kept)
  get_seeds                MinimizerSeeder::getSeeds + addMinimizers + matchToSeedHit, src/MinimizerSeeder.cpp:494-555
  order_seeds_by_chaining  GraphAligner::orderSeedsByChaining, src/GraphAligner.h:233-295
  fragment_order           the sort by seqPos of src/Aligner.cpp:667

The three unstable std::sort calls whose tie order matters go through `std_sort` = the local libstdc++'s own std::sort (tests/stdsort/std_sort_perm.cpp). The minimizer index
(sorted k-mers, start offsets, position lists, maxCount) and the graph arrays are inputs: row A0 has its own independent model (tests/graph_model.py)."""
import bisect

MASK64 = (1 << 64) - 1
_CODE = {ord(c): i for i, c in enumerate("ACGT")}
_CODE.update({ord(c): i for i, c in enumerate("acgt")})


def iterate_kmers(seq, k, w):
    """[(position of the k-mer's last base, k-mer)] as iterateKmers reports them."""
    out = []
    real_window = w - k + 1
    if len(seq) < k:
        return out
    mask = (1 << (2 * k)) - 1
    offset = 0
    while True:   # `start:`
        while offset < len(seq) and seq[offset] not in _CODE:
            offset += 1
        if offset + k > len(seq):
            return out
        kmer, restart = 0, False
        for i in range(k):
            if seq[offset + i] not in _CODE:
                offset += i
                restart = True
                break
            kmer = (kmer << 2) | _CODE[seq[offset + i]]
        if restart:
            continue
        out.append((offset + k - 1, kmer))
        last_kmer, last_pos = kmer, offset + k - 1
        i = k
        while offset + i < len(seq):
            if seq[offset + i] not in _CODE:
                offset += i
                restart = True
                break
            kmer = ((kmer << 2) & mask) | _CODE[seq[offset + i]]
            if last_kmer != kmer or last_pos <= ((offset + i - real_window) & MASK64):   # size_t arithmetic: a small position wraps to a huge one
                out.append((offset + i, kmer))
                last_kmer, last_pos = kmer, offset + i
            i += 1
        if not restart:
            return out


def get_seeds(seq, index, graph, k, w, density, std_sort):
    """MinimizerSeeder::getSeeds: the read's seed hits in the order addMinimizers pushes them. index: kmers (sorted), start, positions, maxcount; graph: nodeIDs, nodeOffset, reverse."""
    kmers, start, positions, max_count = index["kmers"], index["start"], index["positions"], index["maxcount"]
    matches = []                                            # (pos, bucket start, count)
    for pos, kmer in iterate_kmers(seq, k, w):
        at = bisect.bisect_left(kmers, kmer)
        if at == len(kmers) or kmers[at] != kmer:
            continue
        count = start[at + 1] - start[at]
        if count >= max_count:
            continue
        matches.append((pos, start[at], count))
    max_hits = (1 << 64) - 1 if density == -1 else int(len(seq) * density)
    # addMinimizers: prefer the less common minimizers (an unstable sort by count), then take whole position lists until the hit budget is spent -
    # a list as long as the last one taken is still taken
    order = std_sort([m[2] for m in matches])
    seeds, seeds_here, allowed = [], 0, 0
    for m in (matches[i] for i in order):
        pos, first, count = m
        if seeds_here >= max_hits and count > allowed:
            break
        allowed = count
        for i in range(first, first + count):
            merged = positions[i]
            node, offset = merged >> 6, merged & 63
            seeds.append({"nodeID": graph["nodeIDs"][node] // 2, "nodeOffset": offset + graph["nodeOffset"][node], "seqPos": pos, "matchLen": k, "raw": max_count - count,
                          "reverse": bool(graph["reverse"][node]), "agNode": node, "agOffset": offset, "goodness": 0, "cluster": 0})
        seeds_here += count
    return seeds


def order_seeds_by_chaining(seeds, graph, std_sort):
    """orderSeedsByChaining: clusters per chain by diagonal (gaps of at most 100), a cluster's matching base pairs, goodness = those + the raw goodness, best first."""
    by_chain = {}
    for i, s in enumerate(seeds):
        node, real_offset = s["agNode"], s["agOffset"]
        by_chain.setdefault(graph["chainNumber"][node], []).append((i, graph["chainApproxPos"][node] + real_offset - s["seqPos"]))
    for members in by_chain.values():
        members.sort(key=lambda m: m[1])                   # (only which seeds share a cluster and the sorted values matter: any order among equals gives the same clusters)
        cluster_start = 0
        for i in range(1, len(members) + 1):
            if i < len(members) and members[i][1] <= members[i - 1][1] + 100:
                continue
            cluster = sorted(members[cluster_start:i], key=lambda m: seeds[m[0]]["seqPos"])
            matching, last_end = 0, -(1 << 31)
            for idx, _ in cluster:
                this_start = seeds[idx]["seqPos"] - seeds[idx]["matchLen"] + 1
                this_end = seeds[idx]["seqPos"]
                matching += this_end - max(this_start, last_end)
                last_end = this_end
            for idx, _ in cluster:
                seeds[idx]["goodness"] = matching + seeds[idx]["raw"]
                seeds[idx]["cluster"] = i - cluster_start
            cluster_start = i
    order = std_sort([s["goodness"] for s in seeds])       # std::sort by seedGoodness (unstable), then std::reverse
    return [seeds[i] for i in reversed(order)]


def fragment_order(seeds, std_sort):
    """src/Aligner.cpp:667: the seeds sorted by read position (unstable) for the fragment windows."""
    return [seeds[i] for i in std_sort([s["seqPos"] for s in seeds])]
