"""vg::Alignment as the reference writes it (src/vg.proto:52-154), for the protobuf Python runtime: a hand-built SUBSET descriptor - only the fields the path sets -
because the reference's generated module (scripts/vg_pb2.py) does not travel to the GPU box. tests/test_oracle_golden.py checks every field here against
tests/golden/vg_schema.expected.json, the field table tests/golden/make_gam_golden.py took from the reference's own descriptor."""
import json
import os

FIELDS = {
    "Edit": [("from_length", 1, "int32", False, None), ("to_length", 2, "int32", False, None), ("sequence", 3, "string", False, None)],
    "Position": [("node_id", 1, "int64", False, None), ("offset", 2, "int64", False, None), ("is_reverse", 4, "bool", False, None), ("name", 5, "string", False, None)],
    "Mapping": [("position", 1, "message", False, "Position"), ("edit", 2, "message", True, "Edit"), ("rank", 5, "int64", False, None)],
    "Path": [("name", 1, "string", False, None), ("mapping", 2, "message", True, "Mapping"), ("is_circular", 3, "bool", False, None), ("length", 4, "int64", False, None)],
    "Alignment": [("sequence", 1, "string", False, None), ("path", 2, "message", False, "Path"), ("name", 3, "string", False, None),
                  ("score", 6, "int32", False, None), ("query_position", 7, "int32", False, None), ("identity", 16, "double", False, None)],
}
TYPE_NUMBER = {"double": 1, "int64": 3, "int32": 5, "bool": 8, "string": 9, "message": 11}    # FieldDescriptorProto.Type


def alignment_class():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fdp = descriptor_pb2.FileDescriptorProto(name="vg_subset_for_tests.proto", package="vgtest", syntax="proto3")
    for name, fields in FIELDS.items():
        m = fdp.message_type.add(name=name)
        for fname, number, ftype, repeated, type_name in fields:
            f = m.field.add(name=fname, number=number, type=TYPE_NUMBER[ftype], label=F.LABEL_REPEATED if repeated else F.LABEL_OPTIONAL)
            if type_name:
                f.type_name = ".vgtest." + type_name
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fdp)
    desc = pool.FindMessageTypeByName("vgtest.Alignment")
    try:
        return message_factory.GetMessageClass(desc)
    except AttributeError:
        return message_factory.MessageFactory(pool).GetPrototype(desc)


def read_varint(buf, at):
    value, shift = 0, 0
    while True:
        b = buf[at]
        at += 1
        value |= (b & 0x7F) << shift
        if not b & 0x80:
            return value, at
        shift += 7


def decode_gam_stream(raw, Alignment=None):
    """The inflated GAM stream -> [[message dict, ...] per group] (the reader of the reference's scripts/summary.py:63-75: varint count, then varint size + message each),
    every message checked to re-serialise to its own bytes (canonical proto3, nothing outside the descriptor)."""
    from google.protobuf import json_format
    Alignment = Alignment or alignment_class()
    groups, at = [], 0
    while at < len(raw):
        count, at = read_varint(raw, at)
        group = []
        for _ in range(count):
            size, at = read_varint(raw, at)
            msg = Alignment()
            msg.ParseFromString(raw[at:at + size])
            at += size
            assert msg.SerializeToString() == raw[at - size:at]
            through_names = json_format.ParseDict(json_format.MessageToDict(msg, preserving_proto_field_name=True), Alignment())
            assert through_names.SerializeToString() == raw[at - size:at]     # an unknown field would survive the first round trip, not this one
            group.append(json_format.MessageToDict(msg, preserving_proto_field_name=True))
        groups.append(group)
    return groups


def golden_case(name):
    """(reads, expected groups as dicts, expected inflated stream, chained_better) of a case of tests/golden/make_gam_golden.py."""
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    doc = json.load(open(os.path.join(gold, name + ".expected.gam.json")))
    if name == "ref_test":
        reads = [open(os.path.join(gold, "ref_test_read.fa")).read().split("\n")[1].encode()]
        gfa = os.path.join(gold, "ref_test_graph.gfa")
    elif name == "syn20k":
        reads = [l.strip().encode() for l in open(os.path.join(gold, "syn20k.fa")) if l.strip() and not l.startswith(">")]
        gfa = os.path.join(gold, "syn20k.gfa")
    else:
        reads = open(os.path.join(gold, name + ".reads.txt"), "rb").read().split(b"\n")[:-1]
        gfa = os.path.join(gold, "syn20k.gfa")
    assert len(reads) == doc["reads"]
    return gfa, reads, doc["groups"], open(os.path.join(gold, name + ".expected.gam"), "rb").read(), doc["chained_better"]


# ---- the second harness row of SURVEY.md §8(c): an alignment's path spelled through the GFA, as the reference's scripts/summary.py:77-91 does it (restated in
# tests/golden/make_gam_golden.py beside the reference's own descriptor; here over the decoded message dictionaries, for the boxes where the reference is absent)

def load_gfa_segments(filename):
    """Segment sequences by the aligner's node id = order of first appearance on an S or L line (src/GfaGraph.cpp:164-173): see make_gam_golden.py's load_gfa_segments."""
    ids, VL = {}, {}
    for line in open(filename):
        if line[0] == "S":
            i, s = line[1:].strip().split()[:2]
            VL[ids.setdefault(i, len(ids))] = s
        elif line[0] == "L":
            li, _, ri = line[1:].strip().split()[:3]
            ids.setdefault(li, len(ids))
            ids.setdefault(ri, len(ids))
    return VL


def spell_alignment(aln, VL):
    """summary.py's parse_alignment over MessageToDict(aln): whole segments in path order, reverse-complemented when position.is_reverse; plus the part the alignment covers."""
    comp = {"A": "T", "T": "A", "C": "G", "G": "C"}
    seq, rev_cnt = "", 0
    mapping = aln["path"]["mapping"]
    for x in mapping:
        ll = VL[int(x["position"].get("node_id", 0))]
        if x["position"].get("is_reverse"):
            rev_cnt += 1
            seq += "".join(comp[c] for c in ll[::-1])
        else:
            seq += ll
    start = int(mapping[0]["position"].get("offset", 0))
    used = sum(int(e.get("from_length", 0)) for m in mapping for e in m.get("edit", []))
    return {"name": aln["name"].split()[0], "seq": seq, "path_cnt": len(mapping), "revcnt": rev_cnt, "path_bps": len(seq), "aligned_from": start, "aligned_bps": len(seq[start:start + used])}, seq[start:start + used]


def golden_paths(name):
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return json.load(open(os.path.join(gold, name + ".expected.paths.json")))["reads"]
