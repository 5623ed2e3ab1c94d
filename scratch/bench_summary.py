import json, sys
d=json.load(open(sys.argv[1]))
print(d["kernels"])
print("value %.3e blocks/s (single-stream %.3e)  ms/step %.4f  roofline %.1f GB/s frac %.3f kern_ms %.4f"%(d["value"], d.get("value_single_stream",0), d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["kernel_ms"]))
for k,v in d.get("large_batch",{}).items(): print("LARGE %-12s %.3e blk/s kern_ms %.4f %.0f GB/s frac %.3f"%(k, v["blocks_per_s"], v["kernel_ms"], v["achieved_GBps"], v["frac_of_hbm_peak"]))
for k,v in d.get("paths",{}).items(): print("%-14s %.3e blk/s kern_ms %.4f %.0f GB/s frac %.3f"%(k, v["blocks_per_s"], v["kernel_ms"], v["achieved_GBps"], v["frac_of_hbm_peak"]))
if d.get("cpu_baseline"): print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["single_thread_value"])
