import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["parity_check"])
for k,s in d.get("sub_results",{}).items(): print(k, "%.3e"%s["value"], round(s["ms_per_step"],3), s.get("ms_per_table_pass"), round(s["roofline"]["frac"],3), s["roofline"].get("passes_per_launch"), (s["roofline"]["pmc_source"] or "")[-70:])
if "cpu_baseline" in d: print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
print(d.get("hbm_probe",{}).get("frac"))
