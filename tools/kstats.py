"""Per-kernel launch count and average duration from rocprofv3's results database(s): python tools/kstats.py dir ..."""
import glob, sqlite3, sys
for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/**/*_results.db", recursive=True)):
        db = sqlite3.connect(f)
        print(d)
        for name, n, us in db.execute("select name, count(*), avg(end-start)/1e3 from kernels group by name order by 3 desc limit 4"):
            print("   %-60s n=%-4d avg %.1f us" % (name.split("(")[1 if name.startswith("(") else 0][:60] if not name.startswith("void") else name[5:65], n, us))
