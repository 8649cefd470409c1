"""Per-kernel time summary from a rocprofv3 --kernel-trace results database: prof_read.py <dir-or-db>"""
import glob, sqlite3, sys
for d in sys.argv[1:]:
    for db in (glob.glob(d + "/*.db") if not d.endswith(".db") else [d]):
        c = sqlite3.connect(db)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
        disp = [t for t in tabs if 'kernel_dispatch' in t][0]; sym = [t for t in tabs if 'kernel_symbol' in t][0]
        rows = list(c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                              f"from {disp} d join {sym} s on d.kernel_id=s.id group by 1 order by 3 desc"))
        tot = sum(r[2] for r in rows)
        print(f"# {db}: total kernel time {tot/1e6:.3f} ms")
        print("kernel,calls,total_us,avg_us,min_us,max_us,pct")
        for name, n, t, mn, mx in rows[:25]:
            short = name.split('(')[0].replace('(anonymous namespace)::', '').replace('void ', '')[:60]
            print(f"{short},{n},{t/1e3:.1f},{t/n/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},{100*t/tot:.1f}")
