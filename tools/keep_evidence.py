#!/usr/bin/env python3
"""Copy a measurement file into profiles/ ONLY if it is evidence: non-empty, no Python traceback, no "undefined symbol", and -- for
.json / .jsonl -- every line parses.  Anything else leaves the existing profile untouched and exits non-zero, so an end-of-round
script that forgot to rebuild a variant cannot overwrite a good table with its failure (round 5 did: VERDICT weak 7).

    python3 tools/keep_evidence.py SRC profiles/DST [--min-bytes N] [--must-contain TEXT]
"""
import argparse
import json
import os
import shutil
import sys

BAD = ("Traceback (most recent call last)", "undefined symbol", "Segmentation fault", "HIP error", "hipError")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--min-bytes", type=int, default=64)
    ap.add_argument("--must-contain", action="append", default=[])
    a = ap.parse_args()
    why = None
    if not os.path.isfile(a.src):
        why = "missing"
    else:
        data = open(a.src, "rb").read()
        text = data.decode("utf-8", "replace")
        if len(data) < a.min_bytes:
            why = f"only {len(data)} bytes"
        else:
            for b in BAD:
                if b in text:
                    why = f"contains {b!r}"
                    break
        if why is None:
            for m in a.must_contain:
                if m not in text:
                    why = f"lacks {m!r}"
                    break
        if why is None and a.src.endswith((".json", ".jsonl")):
            try:
                rows = [json.loads(l) for l in text.splitlines() if l.strip()]
                if not rows:
                    why = "no JSON rows"
            except ValueError as e:
                why = f"bad JSON: {e}"
    if why is not None:
        sys.stderr.write(f"keep_evidence: NOT kept {a.src} -> {a.dst}: {why}\n")
        sys.exit(1)
    os.makedirs(os.path.dirname(os.path.abspath(a.dst)), exist_ok=True)
    shutil.copyfile(a.src, a.dst)
    print(f"kept {a.dst} ({len(data)} bytes)")


if __name__ == "__main__":
    main()
