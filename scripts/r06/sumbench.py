import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]; print(d["ms_per_step"], d["value"], "roofline", r["frac"], r["traffic"], r["launch_ms"], [(k["stage"][:12], round(k["launch_ms"]*1e3,1)) for k in r["kernels"]])
s=d["secondary"]
for k in ("config2_batch8192","deepfm_criteo_b8192","din_taobao_b8192","mmoe_zch_b8192","sharded_w1_proxy_b8192"):
    v=s[k]; print(k, {kk:(round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("ms_per_step","graph_ms_per_step")})
print(d["e2e"]["ms_per_step"], d["cpu_baseline"]["value"], d.get("settle_steps"))
