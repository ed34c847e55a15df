import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
print(sys.argv[1], "value %.3e ms/step %.3f latency %.3f"%(d["value"],d["ms_per_step"],d["latency_ms_single_msm"]))
print("  alone:", {k:round(v,3) for k,v in d["stage_ms"].items() if k!="note"})
print("  piped:", {k:round(v,3) for k,v in d["stage_ms_in_timed_region"].items() if k!="note"})
