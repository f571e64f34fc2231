"""Extracts the two fixed 250-bp input vectors of the reference's `test_sample`
(/root/reference/src/msp.rs:551-581; inputs only -- the reference prints, asserts nothing)
into msp_sample_vectors.json.  Run in the build container (needs /root/reference)."""
import json, re, os
src = open("/root/reference/src/msp.rs").read().split("fn test_sample")[1]
vecs = []
for name in ("v1", "v2"):
    m = re.search(r"let %s: Vec<u8> = vec!\[(.*?)\];" % name, src, re.S)
    vecs.append([int(x) for x in re.findall(r"\d+", m.group(1))])
assert all(len(v) == 250 for v in vecs), [len(v) for v in vecs]
json.dump({"k": 35, "p": 5, "v1": vecs[0], "v2": vecs[1]},
          open(os.path.join(os.path.dirname(__file__), "msp_sample_vectors.json"), "w"))
