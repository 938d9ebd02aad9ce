import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for i, (ra, rb) in enumerate(zip(a, b)):
    line = f"ffn call {i} N {ra['h'].shape[0]}:"
    for k in ("h", "gate", "res", "out", "g_out", "g_h", "g_gate"):
        if ra.get(k) is None:
            continue
        d = (ra[k].double() - rb[k].double()).norm() / (rb[k].double().norm() + 1e-30)
        line += f"  {k} {float(d):.1e}"
    print(line)
    if "g_h" in ra:
        # where does g_h differ?  per coefficient row
        d = (ra["g_h"].double() - rb["g_h"].double())
        per_k = d.pow(2).sum((0, 2)).sqrt() / (rb["g_h"].double().pow(2).sum((0, 2)).sqrt() + 1e-30)
        print("      g_h rel diff per coefficient row:", " ".join(f"{float(v):.0e}" for v in per_k))
