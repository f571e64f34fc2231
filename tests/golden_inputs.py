"""Fixed input vectors held by the reference's own tests (data only)."""
import json
import os

_d = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "msp_sample_vectors.json")))
MSP_V1, MSP_V2 = _d["v1"], _d["v2"]
