import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
print("loss", a["loss"], b["loss"])
for k in ("xa", "xl", "ga", "gl"):
    d = (a[k].double() - b[k].double()).norm() / (b[k].double().norm() + 1e-30)
    print(k, f"rel diff {float(d):.2e}", "max abs", float((a[k] - b[k]).abs().max()))
for i, (ma, mb) in enumerate(zip(a["masks"], b["masks"])):
    print(f"pos_ffn {i}: {int((ma != mb).sum())} of {ma.numel()} ReLU gates differ")
