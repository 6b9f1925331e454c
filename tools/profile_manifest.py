"""Records which kernel sources the rocprofv3 summaries under profiles/ were taken from (bench.py compares: roofline.profiles_match_csrc).
usage: python tools/profile_manifest.py <file under profiles/> ...   (run right after copying new summaries into profiles/)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
mp = os.path.join(ROOT, "profiles", "manifest.json")
m = json.load(open(mp)) if os.path.exists(mp) else {}
m["csrc_sha16"] = bench.csrc_digest()
m["files"] = sorted(set(m.get("files", [])) | set(os.path.basename(f) for f in sys.argv[1:]))
json.dump(m, open(mp, "w"), indent=1)
print(m)
