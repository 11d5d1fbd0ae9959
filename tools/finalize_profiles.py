"""After tools/refresh_profiles.sh + profiles/run_profile.sh + a plain `python bench.py > gpurun_out/bench_final.log` on the GPU box: copies the kept summaries into
profiles/, runs tools/collect_profiles.py, stores the bench line as profiles/<tag>_bench.json and rewrites the hash / headline references in DESIGN.md and
profiles/README.md.   python tools/finalize_profiles.py [tag=r06]"""
import csv, glob, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "prof_" + tag, "keep", "*")):
    shutil.copy(f, os.path.join(ROOT, "profiles"))
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "collect_profiles.py"), tag])
b = json.loads([l for l in open(os.path.join(ROOT, "gpurun_out", "bench_final.log")) if l.startswith("{")][-1])
json.dump(b, open(os.path.join(ROOT, "profiles", tag + "_bench.json"), "w"), indent=1, sort_keys=True)
t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
h, commit = b["roofline"]["source_hash"], t["commit"]
assert h == t["source_hash"] == b["roofline"]["traffic_source"]["profiled_source_hash"], (h, t["source_hash"])
avg = None
for r in csv.DictReader(open(os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))):
    if "k_trace<false, true, 0, false, true, false>" in r["Name"]:
        avg = float(r["AverageNs"]) / 1e6
val, ms, frac, launch = b["value"], b["ms_per_step"], b["roofline"]["frac"], b["roofline"]["avg_launch_ms"]
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
s = re.sub(r"\*\*%s: [0-9a-f]{12}, commit [0-9a-f]{7}," % tag, "**%s: %s, commit %s," % (tag, h, commit), s)
s = re.sub(r"\*\*[0-9,]+ Mrays/s, [0-9.]+ ms per frame\*\* on the profiled box", "**%s Mrays/s, %.3f ms per frame** on the profiled box" % (format(int(round(val)), ","), ms), s)
s = re.sub(r"`roofline.frac` [0-9.]+ \(closest [0-9.]+ ms per launch by HIP events, [0-9.]+ in rocprofv3's", "`roofline.frac` %.3f (closest %.3f ms per launch by HIP events, %.3f in rocprofv3's" % (frac, launch, avg), s)
s = re.sub(r"Headline \(one MI355X\): \*\*[0-9,]+ Mrays/s, [0-9.]+ ms per frame\*\*", "Headline (one MI355X): **%s Mrays/s, %.3f ms per frame**" % (format(int(round(val)), ","), ms), s)
open(p, "w").write(s)
p = os.path.join(ROOT, "profiles", "README.md")
s = open(p).read()
s = re.sub(r"value [0-9,]+ Mrays/s", "value %s Mrays/s" % format(int(round(val)), ","), s)
s = re.sub(r"`source_hash` [0-9a-f]{12} = `traffic.json`, commit [0-9a-f]{7}", "`source_hash` %s = `traffic.json`, commit %s" % (h, commit), s)
open(p, "w").write(s)
print("profiles at %s, source hash %s: %.0f Mrays/s, %.4f ms per frame, frac %.3f, closest %.4f ms (rocprofv3 %.4f)" % (commit, h, val, ms, frac, launch, avg))
