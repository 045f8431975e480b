"""Mint tests/golden/ref_gridencoder_gfx950.npz: inputs and outputs of the REFERENCE's own grid encoder (oracle/_ref/
_ref_gridencoder: gridencoder.cu built for gfx950, see oracle/build_ref.py for the recipe and its one respelling) run on
an MI355X.  TEST INFRASTRUCTURE ONLY.  Run on the GPU box:

    gpurun -- 'python -B oracle/mint_ref_grid_gpu.py gpurun_out/ref_gridencoder_gfx950.npz'

and copy the file to tests/golden/.  The CPU suite (tests/test_oracle_vs_ref_kernels_golden.py) then holds the C oracle's
grid_encode_forward / backward to these arrays: forward and Jacobian bit for bit (per level, with the platform's exp2f
named when it is not glibc's), backward to 1e-4 (the kernel scatters with atomics).  Data only.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = "cuda"

# tag -> grid_offsets keywords + (D, C, gridtype, base resolution H, levels L)
CASES = {
    "d3c2hash": (dict(input_dim=3, num_levels=6, level_dim=2, base_resolution=4, log2_hashmap_size=8, desired_resolution=160), 0),
    "d3c1hash": (dict(input_dim=3, num_levels=6, level_dim=1, base_resolution=4, log2_hashmap_size=8, desired_resolution=160), 0),
    "d3c4hash": (dict(input_dim=3, num_levels=6, level_dim=4, base_resolution=4, log2_hashmap_size=8, desired_resolution=160), 0),
    "d3c8tiled": (dict(input_dim=3, num_levels=6, level_dim=8, base_resolution=4, log2_hashmap_size=8, desired_resolution=160), 1),
    "d2c2hash": (dict(input_dim=2, num_levels=6, level_dim=2, base_resolution=4, log2_hashmap_size=8, desired_resolution=160), 0),
    "d2c4tiled": (dict(input_dim=2, num_levels=6, level_dim=4, base_resolution=4, log2_hashmap_size=8, desired_resolution=160), 1),
    # BASELINE's per-level scales (L16, base 16, desired resolution 2048 * bound) on a small table
    "baseline_bound2": (dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=9, desired_resolution=4096), 0),
    "baseline_bound3": (dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=9, desired_resolution=6144), 0),
}


def main(out):
    from oracle import build_ref as br, oracle as O
    ge = br.load("gridencoder")
    z = {}
    for i, (tag, (kw, gridtype)) in enumerate(CASES.items()):
        offsets, pls = O.grid_offsets(**kw)
        D, C, L, Hb = kw["input_dim"], kw["level_dim"], kw["num_levels"], kw["base_resolution"]
        S = float(np.log2(pls))
        B = 192
        rng = np.random.default_rng(900 + i)
        emb = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)
        x = rng.uniform(0, 1, (B, D)).astype(np.float32)
        x[0] = 0.0; x[1] = 1.0; x[2, 0] = 1.25; x[3, 1] = -0.1
        g = rng.normal(size=(L, B, C)).astype(np.float32)
        cx, ce, co = (torch.from_numpy(a).to(DEV) for a in (x, emb, offsets))
        y = torch.empty(L, B, C, device=DEV); jac = torch.empty(B, L * D * C, device=DEV)
        ge.grid_encode_forward(cx, ce, co, y, B, D, C, L, S, Hb, True, jac, gridtype)
        gemb = torch.zeros_like(ce); gin = torch.zeros(B, D, device=DEV)
        ge.grid_encode_backward(torch.from_numpy(g).to(DEV), cx, ce, co, gemb, B, D, C, L, S, Hb, True, jac, gin, gridtype)
        for n, v in (("x", x), ("emb", emb), ("offsets", offsets), ("S", np.float32(S)), ("H", np.int32(Hb)),
                     ("gridtype", np.int32(gridtype)), ("y", y), ("dy_dx", jac), ("g", g), ("grad_emb", gemb),
                     ("grad_x", gin)):
            z[f"{tag}_{n}"] = v.cpu().numpy() if isinstance(v, torch.Tensor) else v
    torch.cuda.synchronize()
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    np.savez_compressed(out, **z)
    print(f"[mint_ref_grid_gpu] {out}: {len(z)} arrays, {os.path.getsize(out) / 1024:.0f} KiB")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ref_gridencoder_gfx950.npz")
