"""Pins the GAM bytes to the REFERENCE'S OWN schema and reader (run in the build container only; the outputs are committed).

What of the reference runs here: `/root/reference/scripts/vg_pb2.py` - the descriptor protoc generated from `src/vg.proto` -
and the GAM reader of `/root/reference/scripts/summary.py:63-75`. Neither travels: this script imports the first from where
it lies and restates the second (twelve lines, cited below), decodes the oracle's GAM with them and writes what they see:

  <case>.expected.gam.json   one JSON document per case: {"groups": [[message, ...], ...]} with every message as
                             MessageToDict(preserving_proto_field_name=True) of the reference's vg_pb2.Alignment
  <case>.expected.gam        the inflated stream itself (groups concatenated), for byte comparisons
  vg_schema.expected.json    name, number, type and label of every field of Alignment / Path / Mapping / Position / Edit in the
                             reference's descriptor: the tests' hand-built subset descriptor (tests/vg_descriptor.py, which is
                             what decodes on the GPU box, where the reference is absent) is checked against this table

Checks made while generating (any failure aborts):
  - every message parses with the reference's descriptor and RE-SERIALISES TO ITS OWN BYTES (no unknown field, no
    non-canonical encoding: what protobuf's C++ SerializeToString would write for the same message, proto3 being
    deterministic for messages without maps);
  - the stream, deflated one gzip member per group as writeGAMToQueue does (src/Aligner.cpp:261-281), reads back through the
    restated summary.py reader (a single gzip.GzipFile over the concatenated members, varint32 count, varint32 sizes);
  - the oracle's JSON lines parse INTO the reference's descriptor (json_format.Parse with the real field names and types) and
    give the same messages as the GAM: JSON and GAM describe the same vg::Alignment objects.
The protobuf runtime must be the pure-Python one (the generated module predates upb descriptors):
    PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION=python python tests/golden/make_gam_golden.py
Cases: the reference's test/graph.gfa + test/read.fa; the synthetic 20 kbp fixture's six reads; the same graph with a
reverse-strand read, a chimeric read, a read carrying a deletion the graph lacks (its chained alignment wins), a read with
IUPAC letters and a read too short to align (no group).
"""
import gzip
import io
import json
import os
import sys

os.environ.setdefault("PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION", "python")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/scripts")

import vg_pb2  # noqa: E402  (the reference's generated module)
from google.protobuf import json_format  # noqa: E402

from oracle import Oracle  # noqa: E402


def _varint_decoder(mask):
    """scripts/summary.py:43-61 (protobuf's own pure-Python decoder, as the reference copied it)."""
    def decode(buf, pos):
        result = 0
        shift = 0
        while True:
            b = buf[pos]
            result |= (b & 0x7F) << shift
            pos += 1
            if not b & 0x80:
                return result & mask, pos
            shift += 7
    return decode


_decode_varint32 = _varint_decoder((1 << 32) - 1)


def read_alignments(fileobj):
    """scripts/summary.py:63-75 restated: one GzipFile over the whole file, groups of (count, (size, Alignment)*)."""
    buf = gzip.GzipFile(fileobj=fileobj).read()
    n = 0
    while n < len(buf):
        an, n = _decode_varint32(buf, n)
        group = []
        for _ in range(an):
            msg_len, n = _decode_varint32(buf, n)
            raw = buf[n:n + msg_len]
            n += msg_len
            aln = vg_pb2.Alignment()
            aln.ParseFromString(raw)
            assert aln.SerializeToString() == raw, "message does not re-serialise to its own bytes"
            # unknown fields survive a parse / serialise round trip; a trip through the field-name dictionary drops them, so this catches them at every depth
            through_names = json_format.ParseDict(json_format.MessageToDict(aln, preserving_proto_field_name=True), vg_pb2.Alignment())
            assert through_names.SerializeToString() == raw, "a field the reference's schema does not know"
            group.append(aln)
        yield group


def revcomp(s):
    return s.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]


def load_gfa_segments(filename):
    """scripts/summary.py:19-33 restated (the S lines; its edge table is never read): segment sequences by node id. The harness keys VL by the S line's integer NAME and
    looks it up by position.node_id, i.e. it expects a GFA whose names ARE the aligner's node ids; the aligner numbers segments in order of first appearance on an S or L
    line (src/GfaGraph.cpp:164-173; numberBackToIntegers is commented out, :324-328) and the fixtures' GFAs count their names from 1, so the key here is that order of
    appearance - what the S line's name would be in a GFA written the way the harness expects."""
    ids = {}
    VL = {}
    for line in open(filename).readlines():
        if line[0] == "S":
            i, s = line[1:].strip().split()[:2]
            VL[ids.setdefault(i, len(ids))] = s
        elif line[0] == "L":
            li, _, ri = line[1:].strip().split()[:3]
            ids.setdefault(li, len(ids))
            ids.setdefault(ri, len(ids))
    return VL


def parse_alignment(aln, VL):
    """scripts/summary.py:77-91 restated: the alignment's path SPELLED through the GFA - every mapping's whole segment, reverse-complemented when
    position.is_reverse - with the counts the reference's harness tabulates. (Its `revc` maps A, C, G, T only, as this does.)"""
    revc = lambda s: "".join({"A": "T", "T": "A", "C": "G", "G": "C"}[c] for c in s[::-1])
    name = aln.name.split()[0]
    seq = ""
    rev_cnt = 0
    for x in aln.path.mapping:
        ll = VL[x.position.node_id]
        if x.position.is_reverse:
            rev_cnt += 1
            seq += revc(ll)
        else:
            seq += ll
    return {"name": name, "seq": seq, "path_cnt": len(aln.path.mapping), "revcnt": rev_cnt, "path_bps": len(seq)}


def aligned_part(aln, spelled):
    """The letters of the spelled path the alignment covers: from the first mapping's offset, as many as its edits consume of the graph (from_length)."""
    start = aln.path.mapping[0].position.offset
    used = sum(e.from_length for m in aln.path.mapping for e in m.edit)
    return spelled[start:start + used]


def cases():
    from graphchainer_amd.synth import SynthGraph
    read = open(os.path.join(HERE, "ref_test_read.fa")).read().split("\n")[1].encode()
    yield "ref_test", os.path.join(HERE, "ref_test_graph.gfa"), [read]
    syn = [l.strip().encode() for l in open(os.path.join(HERE, "syn20k.fa")) if l.strip() and not l.startswith(">")]
    yield "syn20k", os.path.join(HERE, "syn20k.gfa"), syn
    sg = SynthGraph(20_000, seed=7)             # the generator of syn20k.gfa (make_golden.py)
    more = sg.sample_reads(3, 3000, seed=29)
    sv = more[2][:1100] + more[2][1900:]        # 800 bases the graph has and the read lacks: the chained alignment's case
    iupac = bytearray(more[1][:1500])
    for at, c in ((100, b"N"), (101, b"N"), (640, b"R"), (900, b"Y"), (1201, b"K")):
        iupac[at:at + 1] = c
    yield "syn20k_more", os.path.join(HERE, "syn20k.gfa"), [revcomp(syn[0]), syn[1][:700] + syn[3][200:1300], sv, bytes(iupac), b"ACGTACGTAC", more[0]]


def schema_table():
    """The reference descriptor's fields for the five messages the path writes (src/vg.proto:52-154 as protoc compiled it)."""
    table = {}
    for msg in (vg_pb2.Alignment, vg_pb2.Path, vg_pb2.Mapping, vg_pb2.Position, vg_pb2.Edit):
        d = msg.DESCRIPTOR
        table[d.name] = [{"name": f.name, "number": f.number, "type": f.type, "repeated": bool(f.is_repeated) if hasattr(f, "is_repeated") else f._label == 3, "message": f.message_type.name if f.message_type else None} for f in d.fields]
    return {"package": vg_pb2.DESCRIPTOR.package, "syntax": "proto3", "messages": table}


def main():
    with open(os.path.join(HERE, "vg_schema.expected.json"), "w") as f:
        json.dump(schema_table(), f, indent=1, sort_keys=True)
        f.write("\n")
    for name, gfa, reads in cases():
        ora = Oracle(gfa, long_pass=True)
        res = ora.align(reads)
        groups = ora.gam_groups()
        # the file as the reference writes it: one gzip member per group
        blob = b"".join(gzip.compress(g) for g in groups)
        decoded = list(read_alignments(io.BytesIO(blob)))
        assert len(decoded) == len(groups)
        assert gzip.decompress(blob) == b"".join(groups)
        # JSON lines of the same alignments, parsed into the reference's descriptor
        lines = ora.json().decode().splitlines()
        flat = [m for g in decoded for m in g]
        assert len(lines) == len(flat) >= 1, name
        for line, msg in zip(lines, flat):
            from_json = json_format.Parse(line, vg_pb2.Alignment())
            assert from_json == msg and from_json.SerializeToString() == msg.SerializeToString()
        # the second harness row of SURVEY.md §8(c): every alignment's path spelled through the GFA as scripts/summary.py:77-91 does it. What the spelled path is held to:
        # cut to the part the alignment covers, its NW edit distance to the read is the distance the pipeline reports (whole-read alignment: the first of the
        # read's selection, src/Aligner.cpp:376-408; chained alignment: src/Aligner.cpp:845) - the alignment's CONTENT read back through the reference's own reader
        VL = load_gfa_segments(gfa)
        from oracle import Oracle as _O  # noqa: F401
        from oracle.binding import load_oracle_lib
        olib = load_oracle_lib()
        spelled_doc = []
        with_output = [r for r in range(len(reads)) if res["chained_better"][r] or res["read_long_off"][r + 1] > res["read_long_off"][r]]
        assert len(with_output) == len(decoded), name
        for r, group in zip(with_output, decoded):
            rows = []
            for k, aln in enumerate(group):
                p = parse_alignment(aln, VL)
                part = aligned_part(aln, p["seq"])
                a, b = part.encode(), reads[r]
                d = int(olib.gco_edit_distance(a, len(a), b, len(b)))
                rows.append(dict(p, aligned_from=int(aln.path.mapping[0].position.offset), aligned_bps=len(part), nw_distance_to_read=d))
            # (a chimeric read's group holds its alignments in output order; the reported distance is that of the selection's first, src/Aligner.cpp:376-408)
            want = int(res["chain_edit_distance"][r]) if res["chained_better"][r] else int(res["long_edit_distance"][r])
            assert want in [row["nw_distance_to_read"] for row in rows] and (len(rows) > 1 or rows[0]["nw_distance_to_read"] == want), (name, r, want, [row["nw_distance_to_read"] for row in rows])
            spelled_doc.append({"read": r, "reported_distance": want, "alignments": rows})
        with open(os.path.join(HERE, name + ".expected.paths.json"), "w") as f:
            json.dump({"case": name, "what": "scripts/summary.py:77-91 over the reference-decoded GAM: per alignment the spelled path and its counts; nw_distance_to_read = NW edit distance of the part the alignment covers (aligned_from, aligned_bps) to the whole read", "reads": spelled_doc}, f, separators=(",", ":"), sort_keys=True)
            f.write("\n")
        doc = {"case": name, "reads": len(reads), "chained_better": [int(x) for x in res["chained_better"]],
               "groups": [[json_format.MessageToDict(m, preserving_proto_field_name=True) for m in g] for g in decoded]}
        with open(os.path.join(HERE, name + ".expected.gam.json"), "w") as f:
            json.dump(doc, f, separators=(",", ":"), sort_keys=True)
            f.write("\n")
        with open(os.path.join(HERE, name + ".expected.gam"), "wb") as f:
            f.write(b"".join(groups))
        if name == "syn20k_more":               # the other cases' reads are the committed FASTA files
            with open(os.path.join(HERE, name + ".reads.txt"), "wb") as f:
                f.write(b"\n".join(reads) + b"\n")
        print(name, len(reads), "reads", len(groups), "groups", len(flat), "alignments", sum(doc["chained_better"]), "chained winners", len(blob), "bytes deflated")


if __name__ == "__main__":
    main()
