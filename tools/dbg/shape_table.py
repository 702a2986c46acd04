import sys, re, collections
acc = collections.defaultdict(list)
for line in sys.stdin:
    m = re.match(r"conv (\S+) cin=(\S+) cout=(\d+) k=(\d) s=(\d) splitk=(\d+) (\S*)\s*:\s+([\d.]+) us\s+([\d.]+) TF/s", line)
    if m:
        acc[(m.group(1), m.group(2), m.group(3), m.group(4), m.group(5), m.group(6), m.group(7))].append((float(m.group(8)), float(m.group(9))))
for k, v in sorted(acc.items(), key=lambda kv: -sum(t for t, _ in kv[1])):
    print(f"{k[0]:>12s} cin={k[1]:>8s} cout={k[2]:>5s} k{k[3]} s{k[4]} splitk={k[5]:>2s} {k[6]:>9s} x{len(v):3d}  {sum(t for t,_ in v)/len(v):8.1f} us {sum(f for _,f in v)/len(v):7.1f} TF/s  total {sum(t for t,_ in v)/1e3:7.3f} ms")
