"""Write the serialised tile programs of a clip geometry (fwd0/1/2.vdprog) for the C driver
examples/embed_forward.cpp / vd_program_load.  usage: python tools/export_programs.py OUTDIR T H W [x1|x3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_distillation_amd import plan


def export(outdir, T, H, W, x3=False):
    net = plan.plan_network(plan.NetGeometry(T, H, W), ntw=2, ntw0=1, balanced=not x3)      # (what engine.EmbedEngine plans)
    os.makedirs(outdir, exist_ok=True)
    for li, pl in enumerate(net["fwd"]):
        with open(os.path.join(outdir, "fwd%d.vdprog" % li), "wb") as f:
            f.write(plan.export_program(pl))
    return net


if __name__ == "__main__":
    export(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), x3=(len(sys.argv) > 5 and sys.argv[5] == "x3"))
