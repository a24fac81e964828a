import json,sys
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    r=d.get("resident_in_hbm") or {}
    print(d["row"], "| trainer path utt/s", round(d["value"]), "equiv4s", round(d["equiv_4s_utterances_per_s"]), "ms", round(d["ms_per_step"],1), "| peak GB", round(d["peak_memory_GB"],1), "| sizes", d["utterances_per_batch"][:4], "| resident:", {k: round(v,1) for k,v in r.items()})
