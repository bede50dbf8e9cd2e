import numpy as np, torch
from koemorph_amd import synth
from koemorph_amd.engine import Engine
from koemorph_amd.training import Trainer
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for (d, H, T, B) in ((64, 4, 64, 3), (256, 8, 128, 5), (512, 8, 96, 2), (128, 4, 256, 4)):
    params = synth.make_core_params(3, d, T, 256, "trained")
    L = T * 533 + 40
    audio = dev(synth.make_audio(5, B, L)); emo = dev(synth.normal(6, (B, 256))); target = dev(synth.uniform(7, (B, 52), 0, 1))
    res = {}
    for name, opts in (("default", {}), ("no_pack", {"train_no_fe_pack": 1}), ("no_ln_fuse", {"train_no_ln_fuse": 1}), ("no_dy", {"train_no_dy_split": 1}),
                       ("colsum_gemm", {"train_colsum_gemm": 1}), ("no_dma", {"train_no_dma": 1})):
        e = Engine(d_model=d, num_heads=H, mel_sequence_length=T); e.load_state_dict(params); e.finalize()
        for k, v in opts.items(): e.set_option(k, v)
        tr = Trainer(e, max_windows=B, lr=1e-3, dropout=0.1); tr.set_dropout(0.1, seed=5)
        losses = [float(tr.step(audio, emo, target).item()) for _ in range(3)]
        loss = float(tr.forward_backward(audio, emo, target).item())
        res[name] = (losses, loss, tr.flat_grad.cpu().numpy().copy())
    g0 = res["no_dma"][2]
    out = []
    for k, (ls, l, g) in res.items():
        out.append("%s dl=%.1e dg=%.1e" % (k, abs(l - res["no_dma"][1]), np.abs(g - g0).max() / max(np.abs(g0).max(), 1e-30)))
    print((d, H, T, B), "finite", all(np.isfinite(v[1]) for v in res.values()), "; ".join(out))
