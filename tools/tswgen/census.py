"""tools/tswgen/census.py -- static instruction census of the generated loop, per step variant (ring counter c):
python -m tools.tswgen.census [cfg-dict]   e.g.  python -m tools.tswgen.census "dict(sparse=True)"
(a cfg with loop=3 counts the round-3 loop, tools/tswgen/kernel3.py) """
import sys

from . import kernel as K


def census(cfg):
    cfg = dict(cfg)
    if cfg.pop("loop", 2) == 3:
        from tools.experiments.tswgen3 import kernel3
        p = kernel3.build(cfg)
    else:
        p = K.build(cfg)
    blocks, cur = {}, "prologue"
    for ins in p.ins:
        if ins.op == "label" and ins.src[0].startswith(".LS"):
            cur = ins.src[0]
        elif ins.op == "label" and (ins.src[0].startswith(".Lexit") or ins.src[0].startswith(".Lactz")):
            cur = "stubs"
        c = blocks.setdefault(cur, {})
        o = ins.op
        if o in ("label",):
            continue
        if o == "v_pk_fma_f32":
            k = "pkfma"
        elif o.startswith("v_") and ins.is_dpp():
            k = "dpp"
        elif o.startswith("v_"):
            k = "valu"
        elif o.startswith("ds_"):
            k = "lds"
        elif o.startswith("global_"):
            k = "vmem"
        elif o in ("s_waitcnt", "s_nop"):
            k = "wait"
        elif o == "s_barrier":
            k = "bar"
        elif "branch" in o:
            k = "br"
        else:
            k = "salu"
        c[k] = c.get(k, 0) + 1
    return blocks, len(p.ins)


def main():
    cfg = eval(sys.argv[1]) if len(sys.argv) > 1 else {}
    blocks, n = census(cfg)
    keys = ["pkfma", "dpp", "valu", "salu", "br", "lds", "vmem", "wait", "bar"]
    print("%-14s" % "block" + "".join("%7s" % k for k in keys) + "  total")
    tot = {}
    for name, c in blocks.items():
        print("%-14s" % name[:14] + "".join("%7d" % c.get(k, 0) for k in keys) + "  %5d" % sum(c.values()))
        if name.startswith(".LS"):
            for k in keys:
                tot[k] = tot.get(k, 0) + c.get(k, 0)
    print("%-14s" % "mean per step" + "".join("%7.1f" % (tot.get(k, 0) / 24.0) for k in keys) + "  %5.1f" % (sum(tot.values()) / 24.0))
    print("instructions:", n)


if __name__ == "__main__":
    main()
