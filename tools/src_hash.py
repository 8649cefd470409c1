"""Print the hash bench.py stamps counter summaries with: sha256 over the sources the decode kernels are built from
(bench.decode_source_hash).  tools/collect_profiles.sh writes it as the first line of profiles/rNN_pmc_summary.csv; bench.py
drops `roofline.pmc` / `roofline.traffic` when the committed summary was collected on other sources."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
if len(sys.argv) > 1 and sys.argv[1] == "wide":     # the general-shape decoder's sources (tools/pmc_wide.sh)
    print(bench.source_hash(bench.WIDE_SOURCES))
else:
    print(bench.decode_source_hash())
