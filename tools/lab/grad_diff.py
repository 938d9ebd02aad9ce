import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
rows = []
for n in a:
    d = (a[n].double() - b[n].double())
    rows.append((float(d.norm() / (b[n].double().norm() + 1e-30)), n, float(b[n].norm())))
rows.sort(reverse=True)
for r in rows[:14]:
    print("%.2e  %-60s |g| %.3e" % r)
