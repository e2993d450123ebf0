import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import engines, fuzzdata
g=engines.GpuEngine(); o=engines.OracleEngine()
s = fuzzdata.rows(1, 1500, max_len=40)
for d in (" ", "a", ","):
    for n in (-1, 1, 2):
        a=g.split(s,d,n); b=o.split(s,d,n)
        if a!=b:
            print("MISMATCH", repr(d), n, len(a), len(b))
            for k,(x,y) in enumerate(zip(a,b)):
                bad=[i for i in range(len(x)) if x[i]!=y[i]]
                if bad:
                    i=bad[0]; print(" col",k,"row",i,repr(s[i]),"got",repr(x[i]),"exp",repr(y[i]), "nbad",len(bad)); break
            break
