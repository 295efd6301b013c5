#!/usr/bin/env python3
"""Keep the record of a gpurun call that did not end well (killed by the box's process / memory guard, timed out, GPU fault, non-zero exit):
copies the call's verdict (gpurun_out/.last_call.json: status, rc, the guard's message, stdout / stderr tails, timings) and, if given, the
command line and a note into profiles/<round>/incidents/<UTC time>_<slug>.json -- so that the next reader does not have to take a diagnosis
on trust (a 6-rank rehearsal killed by the process guard left no log).  Run it right after the call, before the next one
overwrites .last_call.json:

    python tools/keep_incident.py r06 "6-rank rehearsal inside pytest" --command "gpurun -- python -m pytest ..." [--note "7 holders > limit 6"]
    python tools/keep_incident.py r06 --if-bad ...        # only when the last call's status is not "ok" or its rc is non-zero
"""
import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("round")
    ap.add_argument("what", nargs="?", default="gpurun call")
    ap.add_argument("--command", default=None)
    ap.add_argument("--note", default=None)
    ap.add_argument("--if-bad", action="store_true")
    a = ap.parse_args()
    last = os.path.join(ROOT, "gpurun_out", ".last_call.json")
    if not os.path.exists(last):
        sys.exit("no gpurun_out/.last_call.json")
    rec = json.load(open(last))
    bad = rec.get("status") != "ok" or rec.get("rc") not in (0, None) or any(rec.get(k) for k in ("fault", "proc_limit", "oom", "silence", "gpu_fault"))
    if a.if_bad and not bad:
        print("last call ended well: nothing kept")
        return
    out = {"what": a.what, "command": a.command, "note": a.note, "kept_at_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
           "ended_badly": bad, "gpurun_verdict": rec}
    d = os.path.join(ROOT, "profiles", a.round, "incidents")
    os.makedirs(d, exist_ok=True)
    slug = re.sub(r"[^a-z0-9]+", "_", a.what.lower()).strip("_")[:48] or "call"
    path = os.path.join(d, time.strftime("%Y%m%dT%H%M%SZ", time.gmtime()) + "_" + slug + ".json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("kept", os.path.relpath(path, ROOT))


if __name__ == "__main__":
    main()
