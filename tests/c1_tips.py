"""Config C1 (BASELINE.json configs[0]): the README pipeline of the reference
(/root/reference/README.md:17-39) on a tips-like CSV.  The public tips.csv is not
available offline, so the 244 data lines are synthesised with its schema
(total_bill,tip,sex,smoker,day,time,size; day in {Thur,Fri,Sat,Sun}) from a
seeded generator -- SURVEY.md section 8(d).  `pipeline(engine)` runs the README's
steps on any engine of tests/engines.py; `pandas_pipeline()` is the same with
pandas.Series.str, the mirror BASELINE.json names for this config."""
import random

DAYS = ["Sun", "Mon", "Tues", "Wed", "Thur", "Fri", "Sat"]  # README.md:30


def lines(seed=20240607, rows=244):
    rnd = random.Random(seed)
    out = []
    for _ in range(rows):
        bill = round(rnd.uniform(3.07, 50.81), 2)
        tip = round(max(1.0, bill * rnd.uniform(0.05, 0.3)), 2)
        out.append("%s,%s,%s,%s,%s,%s,%d" % (bill, tip, rnd.choice(["Female", "Male"]), rnd.choice(["No", "Yes"]),
                                             rnd.choice(["Thur", "Fri", "Sat", "Sun"]), rnd.choice(["Dinner", "Lunch"]),
                                             rnd.randint(1, 6)))
    return out


def pipeline(eng, host_lines):
    cols = eng.split(host_lines, ",", -1)  # gpu_lines.split(',')
    day = cols[4]
    for idx, d in enumerate(DAYS):  # README.md:30-32 (literal replace; the tokens hold no regex metacharacters)
        day = eng.replace(day, d, str(idx), -1)
    keys, values = eng.category(cols[4])  # nvcategory.from_strings(gpu_columns[4])
    return {"columns": cols, "day_encoded": day, "keys": keys, "values": values}


def pandas_pipeline(host_lines):
    import pandas as pd

    s = pd.Series(host_lines)
    frame = s.str.split(",", expand=True)
    day = frame[4]
    for idx, d in enumerate(DAYS):
        day = day.str.replace(d, str(idx), regex=False)
    cat = pd.Categorical(frame[4])
    return {"columns": [frame[c].tolist() for c in frame.columns], "day_encoded": day.tolist(),
            "keys": list(cat.categories), "values": cat.codes.tolist()}
