#!/usr/bin/env python3
"""One clock-settled run per kernel family at roofline-relevant sizes, for rocprofv3.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- \
        python scripts/kernel_bench.py --manifest $OUT/manifest.json [--only fq,tails,...]
    python scripts/summarize_profiles.py kernel-table $OUT profiles/r02

Every family: 0.4 s of unrelated torch work first (so the power management has settled BEFORE the first launch of
the measured kernel -- rocprofv3's per-kernel average then contains no ramp-up launches), 3 warm launches, K timed
launches between HIP events.  The manifest lists, per measured kernel, the name pattern rocprofv3 reports, the
ALGORITHMIC bytes (or ops) of one launch per SURVEY.md 8(d), the bounding resource and its peak;
summarize_profiles.py joins it with the rocprofv3 kernel_stats.csv.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'transformer-quantization_amd'), ROOT):
    sys.path.insert(0, p)

import numpy as np
import torch

from quantization import _hip

HBM = 8000.0                 # GB/s
VALU_PACKED = 157.0e3        # GFLOP/s fp32 vector peak (256 CU x 4 SIMD-32 x 2 flop x 2.4 GHz; reached by plain v_fma_f32 --
                             # v_pk_fma_f32 is the same rate per element, v_med3 / v_rndne half: tools/tuning/valu_probe.hip)
MFMA_I8 = 3.944e6            # GOP/s dense int8 MFMA: the MEASURED v_mfma_i32_16x16x64_i8 rate of MI355X_MICROARCH.md (>= 3944 TOPS)

be = _hip.backend()
dev = 'cuda'
rows_out = []


def settle(seconds=0.4):
    a = torch.empty(1 << 26, device=dev)
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        for _ in range(10):
            a.add_(1.0)
        torch.cuda.synchronize()


def run(family, label, pattern, fn, unit_per_launch, bound, reps=30, note='', patterns=None):
    settle()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    peak = {'hbm': HBM, 'valu': VALU_PACKED, 'mfma_i8': MFMA_I8}[bound]
    ach = unit_per_launch / (ms * 1e-3) / 1e9
    rows_out.append({'family': family, 'label': label, 'pattern': pattern, 'bound': bound,
                     'unit_per_launch': unit_per_launch, 'unit': 'bytes' if bound == 'hbm' else 'ops',
                     'peak': peak, 'peak_unit': 'GB/s' if bound == 'hbm' else 'GOP/s',
                     'event_us': round(ms * 1e3, 2), 'event_achieved': round(ach, 1),
                     'event_frac': round(ach / peak, 4), 'note': note, **({'patterns': patterns} if patterns else {})})
    print(f'{family:8s} {label:46s} {ms * 1e3:9.1f} us  {ach:10.1f} {"GB/s" if bound == "hbm" else "GOP/s"}'
          f'  {100 * ach / peak:5.1f} %', flush=True)


def q7(d, z):
    return (torch.tensor(d, device=dev), torch.tensor(z, device=dev), None, 8, False, False, 1e-8)


def fam_fq():
    B, S, D = 1024, 512, 768
    for dt, es, code in ((torch.bfloat16, 2, 1), (torch.float32, 4, 0)):
        x = torch.randn(B, S, D, device=dev).to(dt)
        n = x.numel()
        delta, zf = torch.tensor(0.03, device=dev), torch.tensor(128.0, device=dev)
        dv, zv = torch.full((D,), 0.03, device=dev), torch.full((D,), 128.0, device=dev)
        name = str(dt)[6:]
        run('fq', f'K1 fq_tensor {name} [1024,512,768]', f'fq_tensor<{code}, false, true', lambda: be.fake_quant(
            x, delta, zf, None, 8, False, False, 1e-8, 1, 1), 2 * es * n, 'hbm')
        run('fq', f'K2 fq_axis per-embedding {name} [1024,512,768]', f'fq_axis', lambda: be.fake_quant(
            x, dv, zv, None, 8, False, False, 1e-8, D, 1), 2 * es * n, 'hbm', note=f'dtype code {code}')
        # the [B, T, 3072] feed-forward activations of config 2 (per-embedding ranges on wide rows)
        xw = x.view(B, S // 4, 4 * D)
        dw, zw = torch.full((4 * D,), 0.03, device=dev), torch.full((4 * D,), 128.0, device=dev)
        run('fq', f'K2w fq_axis per-embedding {name} [1024,128,3072]', f'fq_axis', lambda: be.fake_quant(
            xw, dw, zw, None, 8, False, False, 1e-8, 4 * D, 1), 2 * es * n, 'hbm', note=f'dtype code {code}')
        run('fq', f'K3 index-only u8 {name}', f'fq_tensor<{code}, true', lambda: be.fake_quant(
            x, delta, zf, None, 8, False, False, 1e-8, 1, 1, want_y=False, idx_dtype=torch.uint8), (es + 1) * n, 'hbm')
        gy = torch.randn(B, S, D, device=dev).to(dt)
        run('fq', f'STE backward {name}', f'fq_bwd_tensor<{code}', lambda: be.fake_quant_bwd(
            x, gy, delta, zf, None, 8, False, False, 1e-8, 1, 1), 3 * es * n, 'hbm')
        run('stats', f'K4 mm_rows per-tensor {name}', f'mm_rows<{code}', lambda: be.minmax(x, 1, 1), es * n, 'hbm')
        run('stats', f'K5 mm_cols per-embedding {name}', f'mm_cols<{code}', lambda: be.minmax(x, D, 1), es * n, 'hbm')
        del gy
        # per-token ranges (`--per-token`: axis = 1, reference main.py:359-376): one (scale, zero-point) per token position
        dT, zT = torch.full((S,), 0.03, device=dev), torch.full((S,), 128.0, device=dev)
        run('fq', f'K2r fq_rows_tab per-token {name} [1024,512,768]', 'fq_rows_tab', lambda: be.fake_quant(
            x, dT, zT, None, 8, False, False, 1e-8, S, D), 2 * es * n, 'hbm', note=f'dtype code {code}')
        run('stats', f'K4r mm_rows_wave per-token {name}', 'mm_rows_wave', lambda: be.minmax(x, S, D), es * n, 'hbm')
        # dynamic per-token step (`--dynamic --per-token`): statistics -> estimator -> parameters -> quantize, every call;
        # x is read twice (3 x es bytes per element); the row sums the two streaming kernels of the call, the two
        # parameter-sized launches between them (mm_final, calib_update_k) show in the HIP-event column
        run('dyn', f'dynamic per-token estimate+quantize {name} [1024,512,768]', None, lambda: be.calibrate_minmax(
            x, S, D, _hip.EST_CURRENT, None, None, 0.9, 0, None, 8, False, 1e-8, False), 3 * es * n, 'hbm',
            patterns=['mm_rows_wave', 'fq_rows_tab'], note='tq_calibrate_minmax: 4 launches')
        run('dyn', f'dynamic per-tensor estimate+quantize {name} [1024,512,768]', None, lambda: be.calibrate_minmax(
            x, 1, 1, _hip.EST_CURRENT, None, None, 0.9, 0, None, 8, False, 1e-8, False), 3 * es * n, 'hbm',
            patterns=[f'mm_rows<{code}', f'fq_tensor<{code}, false, true'],
            note='tq_calibrate_minmax with one range: statistics, update, quantize (the same kernels as the K4 / K1 rows)')


def fam_dyn_small():
    # `--dynamic --per-token` at the BASELINE batch (B = 8, T = 128): the per-token data of one position (8 rows x d) fits a
    # block's registers -> statistics, estimator, parameters and quantization as ONE launch with one read of x
    for d in (768,):
        for dt, es, code in ((torch.bfloat16, 2, 1), (torch.float32, 4, 0)):
            x = torch.randn(8, 128, d, device=dev).to(dt)
            name = str(dt)[6:]
            run('dyn', f'dynamic per-token ONE PASS {name} [8,128,{d}]', f'calib_rows_onepass_k<{code}', lambda: be.calibrate_minmax(
                x, 128, d, _hip.EST_CURRENT, None, None, 0.9, 0, None, 8, False, 1e-8, False), 2 * es * x.numel(), 'hbm',
                note='launch-latency bound at this size (1.5 / 3 MB); was 4 launches')


def fam_tails():
    for rows, d in ((131072, 768), (131072, 512)):
        for dt, es, code in ((torch.bfloat16, 2, 1), (torch.float32, 4, 0)):
            a = torch.randn(rows, d, device=dev).to(dt)
            r = torch.randn(rows, d, device=dev).to(dt)
            w, b = torch.randn(d, device=dev), torch.randn(d, device=dev)
            q1, q2, q3 = q7(0.05, 120.0), q7(0.06, 128.0), q7(0.03, 128.0)
            name = str(dt)[6:]
            run('tails', f'residual+LayerNorm tail {name} [{rows},{d}]', 'res_ln_quant_k',
                lambda: be.residual_layernorm_quant(a, r, q1, q2, w, b, 1e-12, q3), 3 * es * a.numel(), 'hbm',
                note=f'dtype code {code}, LayerNorm')
            run('tails', f'residual+NoNorm tail {name} [{rows},{d}]', 'res_ln_quant_k',
                lambda: be.residual_layernorm_quant(a, r, q1, q2, w, b, None, q3), 3 * es * a.numel(), 'hbm',
                note=f'dtype code {code}, NoNorm (affine_only)')
            run('tails', f'NoNorm affine+quant {name} [{rows},{d}]', f'fq_affine<{code}',
                lambda: be.affine_fake_quant(a, w, b, *q3), 2 * es * a.numel(), 'hbm')
    qs, qp = q7(0.5, 128.0), q7(0.004, 0.0)
    # [64,...]: 100 MB of traffic, MALL-sized and short (ramp-up / tail are ~15 % of 22 us); [256,...]: 403 MB, HBM-bound
    for B in (64, 256):
        s = torch.randn(B, 12, 128, 128, device=dev)
        mask = torch.zeros(B, 128, device=dev)
        run('tails', f'scores->softmax->probs fp32 [{B},12,128,128]', 'softmax_quant_k',
            lambda: be.scores_softmax_quant(s, mask, 12 * 128, 8.0, qs, qp), 8 * s.numel(), 'hbm')
        del s, mask


def fam_mse():
    # 10 fp32 operations per element and candidate (div, rint, 2 x clamp, sub / mul dequant, sub,
    # fma-accumulate + compare overhead) against the packed-FMA VALU peak
    for shape, C in (((8, 128, 768), 100), ((8, 128, 768), 12800), ((256, 512, 768), 100), ((3072, 768), 100)):
        x = torch.randn(*shape, device=dev)
        tab = torch.tensor(np.stack([np.linspace(0.01, 0.2, C), np.full(C, 100.0), np.zeros(C), np.full(C, 255.0)],
                                    1).astype(np.float32)).to(dev)
        loss = be.zeros_f64((1, C), dev)
        run('mse', f'K7/K8 ordered candidates {list(shape)} C={C}', 'mse_ord_unit_k',
            lambda: be.mse_candidates_ordered(x, tab, loss), 10.0 * x.numel() * C, 'valu', reps=10,
            note='whole call incl. fold / row-sum kernels in the event time; the rocprof row is mse_ord_unit_k alone')


def fam_i8():
    for (M, N, K) in ((1024, 768, 768), (1024, 3072, 768), (1024, 768, 3072), (8192, 3072, 768)):
        x = torch.randint(-128, 127, (M, K), dtype=torch.int8, device=dev)
        w = torch.randint(-127, 127, (N, K), dtype=torch.int8, device=dev)
        rs = be.rowsum_i8(w)
        b = torch.randn(N, device=dev)
        xd, xz = torch.tensor(0.02, device=dev), torch.tensor(117.0, device=dev)
        wd = torch.tensor(0.001, device=dev).reshape(1)
        qo = q7(0.05, 100.0)
        run('i8', f'integer Linear+GELU+quant M={M} N={N} K={K}', 'linear_i8_lds_k',
            lambda: be.linear_i8(x, w, rs, b, (xd, xz, 8, 1e-8), wd, 1e-8, _hip.ACT_GELU, qo, torch.float32),
            2.0 * M * N * K, 'mfma_i8', note=f'M={M} N={N} K={K}')
        if M == 8192 or N == 3072:
            # index-only output (y = NULL): what BERT's intermediate Linear runs inside quantized_bert_ffn
            run('i8', f'integer Linear+GELU+quant INDEX-ONLY M={M} N={N} K={K}', 'linear_i8_lds_k',
                lambda: be.linear_i8(x, w, rs, b, (xd, xz, 8, 1e-8), wd, 1e-8, _hip.ACT_GELU, qo, torch.float32,
                                     want_idx=True, want_y=False),
                2.0 * M * N * K, 'mfma_i8', note=f'M={M} N={N} K={K}, int8 indices only')
            # the same with GELU + quantizer evaluated through the staircase table (csrc/tq_stair.hip; the default of
            # QuantLinear's integer path, options.INT8_ACT_STAIR)
            st = be.act_stair(_hip.ACT_GELU, qo)
            assert st[0][:16].view(torch.float32)[3].item() == 1.0
            run('i8', f'integer Linear+GELU+quant STAIRCASE M={M} N={N} K={K}', 'linear_i8_lds_k',
                lambda: be.linear_i8(x, w, rs, b, (xd, xz, 8, 1e-8), wd, 1e-8, _hip.ACT_GELU, qo, torch.float32, stair=st),
                2.0 * M * N * K, 'mfma_i8', note=f'M={M} N={N} K={K}, staircase epilogue')
            run('i8', f'integer Linear+GELU+quant STAIRCASE INDEX-ONLY M={M} N={N} K={K}', 'linear_i8_lds_k',
                lambda: be.linear_i8(x, w, rs, b, (xd, xz, 8, 1e-8), wd, 1e-8, _hip.ACT_GELU, qo, torch.float32,
                                     want_idx=True, want_y=False, stair=st),
                2.0 * M * N * K, 'mfma_i8', note=f'M={M} N={N} K={K}, staircase epilogue, int8 indices only')
    # MobileBERT shapes (M = 1024 tokens): Linear 512 -> 128 with the residual NoNorm tail in its epilogue, and a whole
    # feed-forward block (128 -> 512 ReLU quant -> 128 + tail) as one launch
    M = 1024
    g = torch.Generator(device=dev).manual_seed(5)
    def i8(shape, lo=-127, hi=127):
        return torch.randint(lo, hi, shape, dtype=torch.int8, device=dev, generator=g)
    xd, xz = torch.tensor(0.02, device=dev), torch.tensor(117.0, device=dev)
    wd = torch.tensor(0.001, device=dev).reshape(1)
    res = torch.randn(M, 128, device=dev)
    nw, nb = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.1
    x5, w5 = i8((M, 512), -128, 127), i8((128, 512))
    rs5 = be.rowsum_i8(w5)
    b5 = torch.randn(128, device=dev)
    qa, qb, qc, qm = q7(0.05, 100.0), q7(0.06, 110.0), q7(0.05, 120.0), q7(0.02, 0.0)
    run('i8', 'integer Linear 512->128 + residual NoNorm tail M=1024', 'linear_i8_lds_k',
        lambda: be.linear_i8_nonorm(x5, w5, rs5, b5, res, nw, nb, (xd, xz, 8, 1e-8), wd, 1e-8, qa, qb, qc, torch.float32,
                                    want_idx=True), 2.0 * M * 128 * 512, 'mfma_i8')
    x1, w1, w2 = i8((M, 128), -128, 127), i8((512, 128)), i8((128, 512))
    rs1, rs2 = be.rowsum_i8(w1), be.rowsum_i8(w2)
    b1 = torch.randn(512, device=dev)
    run('i8', 'feed-forward block 128->512->128 + tail, one launch M=1024', 'ffn_i8_k',
        lambda: be.ffn_i8_nonorm(x1, (xd, xz, 8, 1e-8), w1, rs1, b1, wd, 1e-8, qm, w2, rs2, b5, wd, 1e-8, res, nw, nb,
                                 qa, qb, qc, torch.float32, want_idx=True),
        2.0 * M * 512 * 128 * 2, 'mfma_i8')
    for B, T, H in ((8, 128, 12), (64, 128, 12)):
        qi, ki, vi = (torch.randint(-128, 128, (B, T, H * 64), dtype=torch.int8, device=dev) for _ in range(3))
        mask = torch.zeros(B, T, device=dev)
        P = [q7(0.02, 120.0), q7(0.02, 130.0), q7(0.01, 128.0), q7(0.5, 128.0), q7(0.003, 0.0), q7(0.01, 128.0)]
        run('i8', f'integer attention core B={B} T={T} H={H}', 'attention_i8_k',
            lambda: be.attention_i8(qi, ki, vi, H, mask, 8.0, *P, want_idx=True), 4.0 * B * H * T * T * 64, 'mfma_i8',
            note=f'B={B} T={T} H={H}')


def fam_ada():
    # [3072,768] / [768,768]: BERT-base Linear weights (28 B/elem x 2.36 M = 66 MB of streams: L2/MALL resident);
    # [30522,768]: the word-embedding table, 23.4 M elements = 656 MB of streams per step: genuinely HBM-bound
    for n_out, n_in in ((3072, 768), (768, 768), (30522, 768)):
        w = torch.randn(n_out, n_in, device=dev) * 0.05
        alpha = torch.randn(n_out, n_in, device=dev)
        g = torch.randn(n_out, n_in, device=dev)
        m, v = torch.zeros_like(w), torch.zeros_like(w)
        qargs = (torch.tensor(0.01, device=dev), None, torch.tensor(True, device=dev), 4, True, False, 1e-8, 1, 1)
        n = w.numel()
        note = 'streams exceed the 256 MB MALL: HBM-bound' if n > (1 << 23) else 'tensors L2/MALL resident'
        run('ada', f'K10 adaround soft forward [{n_out},{n_in}]', 'ada_fwd_k',
            lambda: be.adaround_fwd(w, alpha, qargs, _hip.ADA_HARD_SIGMOID, True, None), 12 * n, 'hbm', note=note)
        run('ada', f'K11 backward+regulariser+Adam [{n_out},{n_in}]', 'ada_bwd_adam_k',
            lambda: be.adaround_bwd_adam(w, g, alpha, m, v, qargs, _hip.ADA_HARD_SIGMOID, None, 0.01, 10.0, 1e-3, 0.9,
                                         0.999, 1e-8, 5), 28 * n, 'hbm', note=note)
        del w, alpha, g, m, v
    a, b = torch.randn(64, 128, 3072, device=dev), torch.randn(64, 128, 3072, device=dev)
    run('ada', 'K13 reconstruction loss [64,128,3072]', 'sqdiff', lambda: be.recon_loss(a, b), 8 * a.numel(), 'hbm')


FAMILIES = {'fq': fam_fq, 'dyn_small': fam_dyn_small, 'tails': fam_tails, 'mse': fam_mse, 'i8': fam_i8, 'ada': fam_ada}

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')
    ap.add_argument('--manifest', default='')
    args = ap.parse_args()
    todo = [f for f in args.only.split(',') if f] or list(FAMILIES)
    with torch.no_grad():
        for f in todo:
            FAMILIES[f]()
            torch.cuda.empty_cache()
    if args.manifest:
        with open(args.manifest, 'w') as fh:
            json.dump(rows_out, fh, indent=1)
