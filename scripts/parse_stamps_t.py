import json,sys
t=open(sys.argv[1]).read()
i=t.rindex("sdf_forward {")
d=json.JSONDecoder().raw_decode(t[i+len("sdf_forward "):])[0]
import re
ms=re.findall(r"sdf_forward \{'median_ms': ([0-9.]+)", t)
print(sys.argv[1], "ms", ms, {k:(round(v["mean"]) if isinstance(v,dict) else v) for k,v in d.items() if k in ("lin1","lin5","lin8 rows 1..256 + feature tile","tile_total")})
