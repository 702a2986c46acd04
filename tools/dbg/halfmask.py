import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from conftest import Golden, max_err
from oracle import synth
from azula_amd.nn import MultiheadSelfAttention
torch.set_grad_enabled(False)
g = Golden("g12_blocks_mask_cond_periodic")
sh = {n: tuple(v) for n, v in g.meta["msa_shapes"].items()}
for half in (False, True):
    msa = MultiheadSelfAttention(64, pos_channels=2, attention_heads=4, rope=True)
    msa.load_state_dict(synth.synth_state_dict(sh, 31))
    msa = msa.cuda().eval()
    if half:
        msa = msa.bfloat16()
    x, pos = g["msa_x"].cuda(), g["msa_pos"].cuda()
    yc = msa(x, pos, g["msa_causal"].bool().cuda())
    yn = msa(x, pos)
    print("half", half, "causal vs causal-ref", max_err(yc, g["msa_y_causal"]), "causal vs nomask-ref", max_err(yc, g["msa_y_nomask"]),
          "nomask vs nomask-ref", max_err(yn, g["msa_y_nomask"]))
    print(" per-row err causal:", (yc - g["msa_y_causal"].cuda()).abs().amax(-1)[0].tolist())
