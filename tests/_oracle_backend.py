"""A backend double for CPU-only tests: same method surface as quantization._hip.HipBackend, but
every call is answered by the CPU oracle.  It exists so the host-side logic of the drop-in classes
(state machines, shape bookkeeping, candidate tables, distributed hooks) can be exercised on a box
without a GPU.  It is test infrastructure and is never importable from the product package."""
import numpy as np
import torch

from oracle import tq_oracle as O

EST_CURRENT, EST_ALL, EST_RUNNING = 0, 1, 2
_MODES = {0: 'learned_sigmoid', 1: 'learned_hard_sigmoid', 2: 'sigmoid_temp_decay'}


class OracleBackend:
    name = 'oracle-double'

    def to_device_f32(self, v, like=None):
        if torch.is_tensor(v):
            return v.detach().float()
        return torch.tensor(v, dtype=torch.float64).float()

    @staticmethod
    def _work(x):
        """compute dtype: fp32 for fp32 / bf16 / fp16 storage, float64 stays float64 (--double)"""
        return x if x.dtype == torch.float64 else x.float()

    def _range_tensor(self, v):
        # float64 range TENSORS keep their dtype, python floats become fp32 tensors (quantizers.py:248-250)
        if torch.is_tensor(v) and v.dtype == torch.float64:
            return v.detach()
        return self.to_device_f32(v)

    def _quant(self, x, delta, zero_float, signed, n_bits, symmetric, log_domain, eps, n_params,
               inner):
        sgn = bool(signed.item()) if signed is not None else False
        xf = self._work(x)
        dom = 'log' if log_domain else 'linear'
        if n_params == 1:
            return O.fake_quant(xf, delta.reshape(()), None if zero_float is None else
                                zero_float.reshape(()), n_bits, symmetric, sgn, eps, dom)
        outer = x.numel() // (n_params * inner)
        xv = xf.reshape(outer, n_params, inner)
        d = delta.reshape(1, n_params, 1)
        z = None if zero_float is None else zero_float.reshape(1, n_params, 1)
        idx, y = O.fake_quant(xv, d, z, n_bits, symmetric, sgn, eps, dom)
        return idx.reshape(x.shape), y.reshape(x.shape)

    def fake_quant(self, x, delta, zero_float, signed, n_bits, symmetric, log_domain, eps,
                   n_params, inner, want_y=True, idx_dtype=None):
        idx, y = self._quant(x, delta, zero_float, signed, n_bits, symmetric, log_domain, eps,
                             n_params, inner)
        if x.dtype == torch.float64 and idx_dtype is not None:
            idx_dtype = torch.float64
        return (y.to(x.dtype) if want_y else None,
                idx.to(idx_dtype) if idx_dtype is not None else None)

    def affine_fake_quant(self, x, w, b, delta, zero_float, signed, n_bits, symmetric, log_domain, eps, want_idx=False):
        r = x.float() * w.float() + b.float()
        idx, y = self._quant(r, delta, zero_float, signed, n_bits, symmetric, log_domain, eps, 1, 1)
        if want_idx:
            return y.to(x.dtype), (idx - 128).to(torch.int8)
        return y.to(x.dtype)

    def residual_layernorm_quant(self, dense_out, residual, q_dense, q_sum, ln_weight, ln_bias, ln_eps, q_out,
                                 want_idx=False):
        def q(v, a):
            return v if a is None else self._quant(v, *a, 1, 1)[1]
        u = q(q(dense_out.float(), q_dense) + residual.float(), q_sum)
        if ln_eps is None:
            v = u * ln_weight.float() + ln_bias.float()
        else:
            v = torch.nn.functional.layer_norm(u, (u.shape[-1],), ln_weight.float(), ln_bias.float(), ln_eps)
        if want_idx:
            idx, y = self._quant(v, *q_out, 1, 1)
            return y.to(dense_out.dtype), (idx - 128).to(torch.int8)
        return q(v, q_out).to(dense_out.dtype)

    def embeddings_layernorm_quant(self, word, word_ids, typ, type_ids, pos, pos_ids, q_sum1, q_sum2, ln_weight, ln_bias, ln_eps,
                                   q_out, want_idx=False):
        """BERT's embedding block through the oracle's element chain (an id outside its table raises IndexError, as torch's
        CPU F.embedding does -- the kernel reports it through its flag; LayerNorm statistics in torch's order, like
        residual_layernorm_quant above)"""
        def q(v, a):
            return v if a is None else self._quant(v, *a, 1, 1)[1]
        pick = lambda table, ids: torch.nn.functional.embedding(ids, table.float())
        u = q(q(pick(word, word_ids) + pick(typ, type_ids), q_sum1) + pick(pos, pos_ids), q_sum2)
        v = torch.nn.functional.layer_norm(u, (u.shape[-1],), ln_weight.float(), ln_bias.float(), ln_eps)
        if want_idx:
            idx, y = self._quant(v, *q_out, 1, 1)
            return y, (idx - 128).to(torch.int8)
        return q(v, q_out)

    # ---- integer evaluation of fixed-range layers (oracle/tq_int_oracle.c): the CPU twin of tq_linear_i8_fwd & co ----
    FFN_SHAPES = {(128, 512, 128)}

    @staticmethod
    def _q7(q):
        """7-tuple with scalar tensors -> python scalars (None stays None)"""
        if q is None:
            return None
        d, z, sg, nb, sym, logd, eps = q
        return (float(d), None if z is None else float(z), None if sg is None else bool(sg), nb, sym, logd, eps)

    def fake_quant_int8(self, x, delta, zero_float, n_bits, eps):
        idx, y = self._quant(x, delta, zero_float, None, n_bits, False, False, eps, 1, 1)
        return y.to(x.dtype), (idx - 128).to(torch.int8)

    def quantize_to_int8(self, x, delta, zero_float, signed, n_bits, symmetric, log_domain, eps, n_params, inner,
                         minus_128):
        idx, _ = self._quant(x, delta, zero_float, signed, n_bits, symmetric, log_domain, eps, n_params, inner)
        return (idx - 128 if minus_128 else idx).to(torch.int8)

    def rowsum_i8(self, w_idx):
        return w_idx.to(torch.int32).sum(1, dtype=torch.int32)

    def act_stair(self, activation, q_out, n_bins=None):
        """The staircase of the HIP backend is a table of the exact specification; here the specification itself is
        evaluated (oracle activation code 4: correctly rounded GELU), so the 'table' is only a marker."""
        return ('oracle-stair', 0)

    def linear_i8(self, x_idx, w_idx, w_rowsum, bias, x_q, w_delta, w_eps, activation, q_out, out_dtype, want_idx=False,
                  want_y=True, stair=None):
        from oracle import int_oracle
        if stair is not None and activation == 2:
            activation = 4
        y, yi = int_oracle.linear_i8(x_idx, w_idx, bias, tuple(float(v) for v in x_q), w_delta, w_eps, activation,
                                     self._q7(q_out))
        y = y.to(out_dtype) if want_y else None
        return (y, yi) if want_idx else y

    def linear_i8_nonorm(self, x_idx, w_idx, w_rowsum, bias, residual, nn_w, nn_b, x_q, w_delta, w_eps, q_dense, q_sum,
                         q_out, out_dtype, want_idx=False):
        from oracle import int_oracle
        y, yi = int_oracle.linear_i8(x_idx, w_idx, bias, tuple(float(v) for v in x_q), w_delta, w_eps, 0,
                                     self._q7(q_dense), tail=1 if residual is None else 2, residual=residual,
                                     nn_w=nn_w, nn_b=nn_b, q_t1=self._q7(q_sum), q_t2=self._q7(q_out))
        return (y.to(out_dtype), yi) if want_idx else y.to(out_dtype)

    def linear_i8_nonorm_grouped(self, x_idx, w_idx, w_rowsum, bias, nn_w, nn_b, x_q, w_delta_rows, w_eps, q_dense, q_out,
                                 out_dtype, want_idx=False, n_groups=2):
        """the chains of the grouped launch, one after the other through the oracle"""
        N, G = w_idx.shape[0], int(n_groups)
        ys, yis = [], []
        for g in range(G):
            sl = slice(g * N // G, (g + 1) * N // G)
            y, yi = self.linear_i8_nonorm(x_idx, w_idx[sl], None, None if bias is None else bias[sl], None, nn_w[sl], nn_b[sl],
                                          x_q, w_delta_rows[sl], w_eps, None if q_dense is None else q_dense[g], None,
                                          None if q_out is None else q_out[g], out_dtype, want_idx=True)
            ys.append(y)
            yis.append(yi)
        return (ys, yis) if want_idx else ys

    def ffn_i8_nonorm(self, x_idx, x_q, w1_idx, w1_rowsum, bias1, w1_delta, w1_eps, q_mid, w2_idx, w2_rowsum, bias2, w2_delta,
                      w2_eps, residual, nn_w, nn_b, q_dense, q_sum, q_out, out_dtype, want_idx=False):
        from oracle import int_oracle
        y, yi = int_oracle.ffn_i8(x_idx, tuple(float(v) for v in x_q), w1_idx, bias1, w1_delta, w1_eps, self._q7(q_mid),
                                  w2_idx, bias2, w2_delta, w2_eps, residual, nn_w, nn_b, self._q7(q_dense),
                                  self._q7(q_sum), self._q7(q_out))
        return (y.to(out_dtype), yi) if want_idx else y.to(out_dtype)

    def ffn_chain_i8_nonorm(self, x_idx, x_q, residual, stages, out_dtype, want_idx=False):
        """the blocks of the chain one after the other through the oracle"""
        y, idx, xq = residual, x_idx, x_q
        for g in stages:
            y, idx = self.ffn_i8_nonorm(idx, xq, g['w1_idx'], g['w1_rowsum'], g['bias1'], g['w1_delta'], g['w1_eps'], g['q_mid'],
                                        g['w2_idx'], g['w2_rowsum'], g['bias2'], g['w2_delta'], g['w2_eps'], y, g['nn_w'],
                                        g['nn_b'], g['q_dense'], g['q_sum'], g['q_out'], out_dtype, want_idx=True)
            q = g['q_out']
            xq = (q[0], q[1], q[3], q[6])
        return (y, idx) if want_idx else y

    def attention_i8(self, q_idx, k_idx, v_idx, num_heads, mask, denom, q_q, q_k, q_v, q_scores, q_probs, q_ctx,
                     want_idx=False):
        from oracle import int_oracle
        ctx, ci = int_oracle.attention_i8(q_idx.contiguous(), k_idx.contiguous(), v_idx.contiguous(), num_heads, mask, denom,
                                          *[self._q7(q) for q in (q_q, q_k, q_v, q_scores, q_probs, q_ctx)])
        return (ctx, ci) if want_idx else ctx

    def linear_i8_grouped(self, x_idx, w_idx, w_rowsum, bias, x_q, w_delta_rows, w_eps, activation, q_outs,
                          want_y=False, want_idx=True, out_dtype=torch.float32):
        from oracle import int_oracle
        G, N = len(q_outs), w_idx.shape[0]
        ys, yis = [], []
        for g in range(G):
            sl = slice(g * N // G, (g + 1) * N // G)
            y, yi = int_oracle.linear_i8(x_idx, w_idx[sl], None if bias is None else bias[sl],
                                         tuple(float(v) for v in x_q), w_delta_rows[sl], w_eps, activation,
                                         self._q7(q_outs[g]))
            ys.append(y)
            yis.append(yi)
        return (torch.cat(ys, -1).to(out_dtype) if want_y else None), (torch.cat(yis, -1) if want_idx else None)

    def scores_softmax_quant(self, scores, mask, rows_per_mask, denom, q_scores, q_probs):
        def q(v, a):
            return v if a is None else self._quant(v, *a, 1, 1)[1]
        s = q(scores.float(), q_scores) / denom
        if mask is not None:
            s = s + mask.reshape(mask.shape[0], 1, 1, mask.shape[-1])
        return q(torch.softmax(s, dim=-1), q_probs)

    def fake_quant_bwd(self, x, grad_y, delta, zero_float, signed, n_bits, symmetric, log_domain,
                       eps, n_params, inner, param_grads=False):
        sgn = bool(signed.item()) if signed is not None else False
        with torch.enable_grad():
            return self._bwd(x, grad_y, delta, zero_float, sgn, n_bits, symmetric, eps, param_grads,
                             'log' if log_domain else 'linear')

    def _bwd(self, x, grad_y, delta, zero_float, sgn, n_bits, symmetric, eps, param_grads, scale_domain='linear'):
        _, dx, dd, dz = O.fake_quant_with_grads(self._work(x), delta.detach(), None if zero_float is None
                                                else zero_float.detach(), n_bits, symmetric, sgn, eps,
                                                grad_out=self._work(grad_y).to(self._work(x).dtype),
                                                scale_domain=scale_domain)
        return dx.to(x.dtype), (dd.reshape(-1) if param_grads else None), (
            dz.reshape(-1) if (param_grads and dz is not None) else
            (torch.zeros(delta.numel()) if param_grads else None))

    def minmax(self, x, n_params=1, inner=1):
        xf = self._work(x.detach())
        if n_params == 1:
            return xf.min(), xf.max()
        outer = x.numel() // (n_params * inner)
        v = xf.reshape(outer, n_params, inner).permute(1, 0, 2).reshape(n_params, -1)
        return v.min(-1)[0], v.max(-1)[0]

    # ---- fused calibration step (tq_calibrate_minmax / tq_calibrate_stats + tq_calibrate_apply) ----------
    accepts_cpu = True
    CALIB_MAX_PARAMS = 4096

    def calibrate_stats(self, x, n_params, inner):
        mn, mx = self.minmax(x, n_params, inner)
        return torch.cat([(-mn).reshape(-1), mx.reshape(-1)]).contiguous()

    def calibrate_apply(self, stats, x, n_params, inner, mode, prev_min, prev_max, momentum, n_groups, order,
                        n_bits, symmetric, eps, log_domain, want_y=True, out=None):
        new_min, new_max = -stats[:n_params], stats[n_params:].clone()
        if n_params == 1:
            new_min, new_max = new_min.reshape(()), new_max.reshape(())
        cur_min, cur_max = self.range_update(mode, new_min, new_max, prev_min, prev_max, momentum, n_groups, order)
        signed = zero_float = None
        if symmetric:
            delta, signed = self.set_range_sym(cur_min, cur_max, n_bits, eps, log_domain)
        else:
            delta, zero_float = self.set_range_asym(cur_min, cur_max, n_bits, eps, log_domain)
        if out is not None:          # in-place state (options.INPLACE_CALIBRATION_STATE)
            for dst, src in zip(out, (cur_min, cur_max, delta, zero_float, signed)):
                if dst is not None:
                    dst.copy_(src.reshape(dst.shape))
            cur_min, cur_max, delta, zero_float, signed = out
        y = None
        if want_y:
            y = self._quant(x, delta, zero_float, signed, n_bits, symmetric, log_domain, eps, n_params, inner)[1]
            y = y.to(x.dtype)
        return cur_min, cur_max, delta, zero_float, signed, y

    def calibrate_minmax(self, x, n_params, inner, mode, prev_min, prev_max, momentum, n_groups, order,
                         n_bits, symmetric, eps, log_domain, want_y=True, out=None):
        return self.calibrate_apply(self.calibrate_stats(x, n_params, inner), x, n_params, inner, mode, prev_min,
                                    prev_max, momentum, n_groups, order, n_bits, symmetric, eps, log_domain,
                                    want_y=want_y, out=out)

    def range_update(self, mode, new_min, new_max, cur_min, cur_max, momentum=0.9, n_groups=0,
                     order=None):
        if n_groups:
            gs = new_min.numel() // n_groups
            perm = order if order is not None else torch.arange(new_min.numel())
            m = new_min[perm].view(n_groups, gs).min(-1)[0].repeat_interleave(gs)
            M = new_max[perm].view(n_groups, gs).max(-1)[0].repeat_interleave(gs)
            new_min, new_max = torch.empty_like(m), torch.empty_like(M)
            new_min[perm], new_max[perm] = m, M
        if mode == EST_CURRENT or cur_min is None:
            return new_min.clone(), new_max.clone()
        if mode == EST_ALL:
            return O.allminmax_update(cur_min, cur_max, new_min, new_max)
        return O.running_update(cur_min, cur_max, new_min, new_max, momentum)

    def axis_ranges(self, new_min, new_max, first):
        r = new_max - new_min
        return r if first else 0.1 * r + (1 - 0.1) * r

    def argsort(self, v):
        return torch.argsort(v)

    def order_stats(self, rows2d, ranks):
        """test double of tq_order_stats: a full sort on the host (the kernel selects without sorting)"""
        srt, _ = torch.sort(rows2d.detach().float(), dim=-1)
        return torch.stack([srt[:, int(r)] for r in ranks], dim=1)

    def set_range_asym(self, x_min, x_max, n_bits, eps, log_domain):
        return O.asym_params_from_range(self._range_tensor(x_min), self._range_tensor(x_max), n_bits,
                                        eps, 'log' if log_domain else 'linear')

    def set_range_sym(self, x_min, x_max, n_bits, eps, log_domain):
        return O.sym_params_from_range(self._range_tensor(x_min), self._range_tensor(x_max), n_bits,
                                       eps, 'log' if log_domain else 'linear')

    # ---- searches: direct evaluation of the (scale, zp, lo, hi) table in fp64 ----------------
    @staticmethod
    def _table_quant(x, c):
        s, zp, lo, hi = (float(v) for v in c)
        s = torch.tensor(s, dtype=torch.float32)
        xi = torch.clamp(torch.round(x / s) + zp, lo, hi)
        return s * (xi - zp)

    def mse_candidates(self, x, rows, cand, loss):
        # fp32 sums in the reference's own order (range_estimators.py:250-256) so that the
        # scipy-driven searches see bit-identical loss values in the CPU tests
        xf = x.detach().float()
        for ci in range(cand.shape[0]):
            y = self._table_quant(xf, cand[ci])
            per_row = torch.sum(((xf - y) ** 2).view(len(xf), -1), dim=1)
            if rows == 1:
                loss[0, ci] += float(torch.sum(per_row))
            else:
                loss[:, ci] += per_row.double()
        return loss

    def mse_candidates_ordered(self, x, cand, loss=None, per_row=False, want_f32=False):
        # literally the reference's two torch.sum calls (range_estimators.py:250-256)
        xf = self._work(x.detach())
        rows = xf.shape[0] if xf.dim() > 0 else 1
        f32 = torch.zeros((rows if per_row else 1, cand.shape[0]), dtype=torch.float32) if want_f32 else None
        if xf.numel() == 0:
            return loss, f32
        for ci in range(cand.shape[0]):
            y = self._table_quant(xf, cand[ci])
            v = torch.sum(((xf - y) ** 2).view(rows, -1), dim=1)
            if not per_row:
                v = torch.sum(v).reshape(1)
            if loss is not None:
                loss[:, ci] += v.double()
            if f32 is not None:
                f32[:, ci] = v
        return loss, f32

    def mse_candidates_grouped(self, x, n_groups, cand, loss):
        # the [n_groups, -1] view the reference's per-channel estimator would see
        xf = x.detach().float()
        xg = xf.transpose(0, xf.dim() - 1).contiguous().view(xf.shape[-1], -1).view(n_groups, -1)
        return self.mse_candidates(xg, n_groups, cand, loss)

    def xent_candidates(self, x, cand, loss):
        if x.numel() == 0:
            return loss
        xf = x.detach().float().reshape(x.shape[0], -1)
        for ci in range(cand.shape[0]):
            y = self._table_quant(xf, cand[ci])
            v = torch.sum(-torch.softmax(xf, 1) * torch.log_softmax(y, 1))
            loss[0, ci] += v.double()
        return loss

    def argmin_select(self, loss, thr_min, thr_max):
        best = torch.from_numpy(np.argmin(loss.numpy(), axis=1))
        return thr_min[best].clone(), thr_max[best].clone(), best

    def candidate_table(self, table_np, device):
        return torch.from_numpy(np.ascontiguousarray(table_np))

    def zeros_f64(self, shape, device):
        return torch.zeros(shape, dtype=torch.float64)

    # ---- AdaRound ---------------------------------------------------------------------------
    def _ada_args(self, w, qargs):
        delta, zf, signed, n_bits, symmetric, log_domain, eps, n_params, inner = qargs
        sgn = bool(signed.item()) if signed is not None else False
        if n_params > 1:
            delta = delta.reshape([-1] + [1] * (w.dim() - 1))
            zf = None if zf is None else zf.reshape([-1] + [1] * (w.dim() - 1))
        return delta, zf, n_bits, symmetric, sgn, eps

    def adaround_fwd(self, w, alpha, qargs, mode, soft, temperature):
        delta, zf, n_bits, symmetric, sgn, eps = self._ada_args(w, qargs)
        return O.ada_fake_quant(w.detach(), alpha.detach(), delta, zf, n_bits, symmetric, sgn,
                                _MODES[mode], soft, eps, temperature)[1]

    def adaround_init_alpha(self, w, qargs, mode, temperature):
        delta, zf, n_bits, symmetric, sgn, eps = self._ada_args(w, qargs)
        return O.ada_alpha_init(w.detach(), O.effective_scale(delta, eps), _MODES[mode], temperature)

    def adaround_bwd(self, w, alpha, grad_wq, qargs, mode, temperature):
        delta, zf, n_bits, symmetric, sgn, eps = self._ada_args(w, qargs)
        with torch.enable_grad():
            a = alpha.detach().clone().requires_grad_(True)
            _, wq = O.ada_fake_quant(w.detach(), a, delta, zf, n_bits, symmetric, sgn, _MODES[mode],
                                     True, eps, temperature)
            wq.backward(grad_wq)
        return a.grad

    def adaround_bwd_adam(self, w, grad_wq, alpha, exp_avg, exp_avg_sq, qargs, mode, temperature,
                          reg_weight, beta, lr, b1, b2, adam_eps, step, want_grad=False):
        delta, zf, n_bits, symmetric, sgn, eps = self._ada_args(w, qargs)
        with torch.enable_grad():
            a = alpha.detach().clone().requires_grad_(True)
            _, wq = O.ada_fake_quant(w.detach(), a, delta, zf, n_bits, symmetric, sgn, _MODES[mode],
                                     True, eps, temperature)
            obj = (wq * grad_wq).sum()
            if reg_weight:
                obj = obj + O.ada_round_reg(a, _MODES[mode], beta, reg_weight, temperature)
            obj.backward()
        g = a.grad
        exp_avg.lerp_(g, 1 - b1)
        exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = (exp_avg_sq.sqrt() / (bc2 ** 0.5)).add_(adam_eps)
        alpha.addcdiv_(exp_avg, denom, value=-(lr / bc1))
        return g if want_grad else None

    def adaround_reg(self, alpha, mode, temperature, beta, weight):
        return O.ada_round_reg(alpha.detach(), _MODES[mode], beta, weight, temperature).double()

    def recon_loss(self, pred, tgt):
        if pred.dim() == 2 and pred.shape[1] == 1:
            # the "plain mean" use of the kernel (LayerOutputMSE): the reference calls F.mse_loss
            return torch.nn.functional.mse_loss(pred.detach().float().reshape(-1), tgt.detach().float().reshape(-1)).double()
        return O.ada_rec_loss(pred.detach().float(), tgt.detach().float()).double()
