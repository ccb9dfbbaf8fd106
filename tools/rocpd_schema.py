import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for name in ("counters_collection", "pmc_events", "pmc_info"):
    try:
        cols = [r[1] for r in db.execute(f"pragma table_info({name})")]
        print(name, cols)
        for r in db.execute(f"select * from {name} limit 2"):
            print("   ", r)
    except Exception as e:
        print(name, "ERR", e)
