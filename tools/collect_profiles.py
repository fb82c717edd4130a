#!/usr/bin/env python3
"""Here, after tools/profile_rNN.sh ran on the GPU box: copy its results into profiles/rNN and write
traffic.json (HBM bytes per launch of the headline kernel from the two counter passes, tied to the
sha256 of the library they were collected on: the one the profiling script wrote down on the box,
which has to be the library in this tree).  usage: tools/collect_profiles.py [r03]"""
import hashlib
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", "prof_" + RND)
dst = os.path.join(ROOT, "profiles", RND)
lib = os.path.join(ROOT, "parquet-go_amd", "libpqdecode.so")
sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()
sha_file = os.path.join(src, "library.sha256")
if os.path.exists(sha_file):
    ran_on = open(sha_file).read().strip()
    if ran_on != sha:
        sys.exit("the counters were collected on library %s, this tree holds %s: profile again" % (ran_on[:12], sha[:12]))
elif RND != "r02":
    sys.exit("no library.sha256 among the results: not collected by tools/profile_%s.sh" % RND)
os.makedirs(dst, exist_ok=True)
c3 = src + "_c3"
if os.path.isdir(c3):  # the all-width sweep, taken in a call of its own
    if open(os.path.join(c3, "library.sha256")).read().strip() != sha:
        sys.exit("the width sweep was taken on another build of the library: take it again")
    shutil.copy(os.path.join(c3, "c3_sweep.jsonl"), os.path.join(src, "c3_sweep.jsonl"))
for name in sorted(os.listdir(src)):
    p = os.path.join(src, name)
    if os.path.isfile(p) and not name.endswith((".err", ".log")):
        shutil.copy(p, os.path.join(dst, name))


def counter(path, name):
    m = re.search(r"%s\s+mean/dispatch = ([0-9.e+]+)" % name, open(path).read())
    return float(m.group(1)) if m else None


fetch = counter(os.path.join(dst, "pmc_fetch.txt"), "FETCH_SIZE")
write = counter(os.path.join(dst, "pmc_write.txt"), "WRITE_SIZE")
rec = {"pages": 5000, "values_per_page": 20000, "width": 16, "kernel": "pq::delta_wave_kernel<unsigned long>",
       "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
       "correction": "FETCH_SIZE x2 for 16-byte-per-lane coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section); "
                     "WRITE_SIZE exact",
       "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024) if fetch and write else None,
       "library_sha256": sha,
       "source": "profiles/%s/pmc_fetch.txt, profiles/%s/pmc_write.txt (separate --pmc passes, tools/profile_%s.sh; "
                 "library.sha256 written on the GPU box at profiling time)" % (RND, RND, RND)}
json.dump(rec, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(rec))
sys.exit(0 if rec["hbm_bytes_per_launch"] else 1)
