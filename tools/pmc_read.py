"""Print per-kernel mean counter values from rocprofv3 --pmc result databases: pmc_read.py <dir>..."""
import glob, sqlite3, sys
for d in sys.argv[1:]:
    for db in glob.glob(d + "/*.db"):
        c = sqlite3.connect(db)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
        pmc = [t for t in tabs if 'pmc_event' in t][0]; info = [t for t in tabs if 'info_pmc' in t][0]
        disp = [t for t in tabs if 'kernel_dispatch' in t][0]; sym = [t for t in tabs if 'kernel_symbol' in t][0]
        q = f"""select s.kernel_name, i.name, count(*), avg(e.value), avg(d.end-d.start) from {pmc} e join {info} i on e.pmc_id=i.id
                join {disp} d on e.event_id=d.event_id join {sym} s on d.kernel_id=s.id group by 1,2"""
        for name, ctr, n, mean, dur in c.execute(q):
            if any(k in name for k in sys.argv[1:] if False) or 'decode_fwd' in name or 'fusion' in name.lower():
                short = name.split('(')[0].split('::')[-1][:40]
                print(f"{short:40s} {ctr:32s} n={n:4d} mean={mean:14.1f} dur_us={dur/1e3:8.1f}")
