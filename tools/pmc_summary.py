"""Summarise rocprofv3 --pmc result databases as CSV rows `key,counter,launches,mean_per_launch`
(a launch's value = the sum over the counter's hardware instances):
pmc_summary.py key=dir [key=dir ...] [--kernel pat1,pat2]   (SQL LIKE patterns without the outer %, any may match)"""
import glob, sqlite3, sys
kern = "decode_fwd"
args = [a for a in sys.argv[1:] if "=" in a]
if "--kernel" in sys.argv:
    kern = sys.argv[sys.argv.index("--kernel") + 1]
print("kernel,counter,launches,mean_per_launch")
for a in args:
    key, d = a.split("=", 1)
    for db in glob.glob(d + "/*.db"):
        c = sqlite3.connect(db)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
        pmc = [t for t in tabs if 'pmc_event' in t][0]; info = [t for t in tabs if 'info_pmc' in t][0]
        disp = [t for t in tabs if 'kernel_dispatch' in t][0]; sym = [t for t in tabs if 'kernel_symbol' in t][0]
        q = f"""select i.name, d.id, sum(e.value) from {pmc} e join {info} i on e.pmc_id=i.id
                join {disp} d on e.event_id=d.event_id join {sym} s on d.kernel_id=s.id
                where {" or ".join(f"s.kernel_name like '%{k}%'" for k in kern.split(","))} group by 1,2"""
        acc = {}
        for name, did, v in c.execute(q):
            acc.setdefault(name, []).append(v)
        for name, vs in sorted(acc.items()):
            print(f"{key},{name},{len(vs)},{sum(vs)/len(vs):.1f}")
