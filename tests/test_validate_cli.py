"""(f4) validate_quantized: flag surface of the reference's quantization options + end-to-end runs on a
2-layer random-init BERT with synthetic tokens."""
import json
import os

import pytest
import torch

import validate_quantized as V


def _cfg(*flags):
    return V.make_config(V.build_parser().parse_args(list(flags)))


def test_flag_surface_and_defaults():
    c = _cfg('--qmethod', 'symmetric_uniform')
    assert c.quant.qmethod_act == 'symmetric_uniform' and c.quant.n_bits == 8 and c.quant.n_bits_act is None
    assert c.quant.act_quant and c.quant.weight_quant and c.quant.quant_setup == 'all'
    assert c.act_quant.quant_method == 'running_minmax' and c.act_quant.num_batches == 1 and c.act_quant.options == {}
    assert c.adaround.layers is None and c.adaround.iters == 10000 or c.adaround.iters > 0
    qp = V.make_qparams(c)
    from quantization.quantizers import QMethods
    from quantization.range_estimators import RangeEstimators
    assert qp['method'] is QMethods.symmetric_uniform and qp['act_range_method'] is RangeEstimators.running_minmax
    assert qp['weight_range_options'] == {} and qp['per_channel_weights'] is False

    c = _cfg('--qmethod', 'symmetric_uniform', '--qmethod-act', 'asymmetric_uniform', '--no-act-quant', '--per-channel',
             '--weight-quant-method', 'MSE', '--weight-opt-method', 'golden_section', '--num-candidates', '50',
             '--act-quant-method', 'MSE', '--act-num-candidates', '64', '--act-opt-method', 'golden_section',
             '--percentile', '99.9', '--quant-dict', "{'y': 'ngp6', 'h': 16}", '--per-token',
             '--adaround', 'layers.0.output.dense, classifier', '--adaround-annealing', '20', '2',
             '--adaround-no-act-func', '--no-adaround-asym')
    assert not c.quant.act_quant and c.quant.per_channel and c.quant.dynamic            # per-token implies dynamic
    assert c.quant.quant_dict == {'y': 'ngp6', 'h': 16}
    assert c.adaround.layers == ('layers.0.output.dense', 'classifier') and c.adaround.annealing == (20.0, 2.0)
    assert c.adaround.include_act_func is False and c.adaround.asym is False
    qp = V.make_qparams(c)
    from quantization.range_estimators import OptMethod
    assert qp['weight_range_options'] == {'opt_method': OptMethod.golden_section, 'num_candidates': 50}
    assert qp['act_range_options'] == {'num_candidates': 64, 'opt_method': OptMethod.golden_section, 'percentile': 99.9}


def test_flag_cross_checks_match_the_reference():
    with pytest.raises(ValueError, match='num_candidates'):
        _cfg('--qmethod', 'symmetric_uniform', '--act-num-candidates', '10')            # only valid with MSE
    with pytest.raises(ValueError, match='momentum'):
        _cfg('--qmethod', 'symmetric_uniform', '--act-quant-method', 'MSE', '--act-momentum', '0.1')
    assert _cfg('--qmethod', 'symmetric_uniform', '--double').double is True
    with pytest.raises(ValueError, match='double'):
        _cfg('--qmethod', 'symmetric_uniform', '--double', '--fast-inference')            # fp32-only kernels
    with pytest.raises(SystemExit):
        V.build_parser().parse_args([])                                                 # --qmethod is required


@pytest.mark.gpu
@pytest.mark.parametrize('flags', [
    ['--qmethod', 'symmetric_uniform', '--qmethod-act', 'asymmetric_uniform', '--num-est-batches', '2'],
    ['--qmethod', 'symmetric_uniform', '--qmethod-act', 'asymmetric_uniform', '--per-groups', '6', '--per-groups-permute',
     '--quant-setup', 'FP_logits', '--act-quant-method', 'current_minmax'],
    ['--qmethod', 'symmetric_uniform', '--qmethod-act', 'asymmetric_uniform', '--per-embd', '--n-bits-act', '6',
     '--quant-dict', "{'h': 16, 'Et': 4, 's': 'fp32'}", '--quant-setup', 'MSE_logits'],
    ['--qmethod', 'symmetric_uniform', '--qmethod-act', 'asymmetric_uniform', '--n-bits', '4', '--n-bits-act', '8',
     '--weight-quant-method', 'MSE', '--adaround', 'layers.0.output.dense', '--adaround-iters', '60',
     '--adaround-num-samples', '16'],
    ['--qmethod', 'symmetric_uniform', '--qmethod-act', 'asymmetric_uniform', '--n-bits', '4', '--n-bits-act', '8',
     '--adaround', 'all', '--adaround-iters', '12', '--adaround-num-samples', '16'],
    ['--qmethod', 'symmetric_uniform', '--qmethod-act', 'asymmetric_uniform', '--num-est-batches', '2', '--double'],
    ['--qmethod', 'symmetric_uniform', '--qmethod-act', 'asymmetric_uniform', '--per-embd', '--double',
     '--weight-quant-method', 'MSE', '--weight-opt-method', 'golden_section', '--act-quant-method', 'current_minmax'],
    # `--per-token` (axis = 1 on every [B, T, d] site, reference main.py:359-376) implies `--dynamic` (main.py:249: no range
    # estimation pass, ranges follow every batch): estimate + quantize on every inference call (mm_rows_wave / fq_rows_wave)
    ['--qmethod', 'symmetric_uniform', '--qmethod-act', 'asymmetric_uniform', '--per-token', '--act-quant-method',
     'current_minmax'],
], ids=['w8a8', 'peg6-permute-fp-logits', 'per-embd-mixed-precision', 'w4a8-adaround', 'w4a8-adaround-all-layers',
        'w8a8-double', 'per-embd-mse-double', 'per-token-dynamic'])
def test_end_to_end_on_two_layers(flags, tmp_path, capsys):
    layers = '1' if 'all' in flags else '2'      # 'all' also walks embeddings (incl. the [1, T] position ids) and LayerNorms
    rep = V.main(flags + ['--num-layers', layers, '--num-eval-batches', '2', '--output-dir', str(tmp_path)])
    assert json.loads(capsys.readouterr().out.strip().splitlines()[-1])['quantizers'] == rep['quantizers']
    f = rep['fidelity_vs_fp32']
    # sanity only: random-init logits are tiny, so the low-bit configurations (W4, 6-bit activations) sit near 0 dB
    assert f['samples'] == 16 and f['logit_sqnr_db'] > (3.0 if not ({'--quant-dict', '--adaround'} & set(flags)) else -10.0), f
    sd = torch.load(os.path.join(tmp_path, 'state_dict.pth'))
    if '--double' in flags:
        assert f['logit_sqnr_db'] > 3.0
        assert all(v.dtype == torch.float64 for k, v in sd.items() if k.endswith('activation_quantizer.quantizer._delta'))
    assert any(k.endswith('activation_quantizer.quantizer._delta') for k in sd)
    if '--per-token' in flags:
        assert rep['timings_s'].get('range_estimation') is None            # dynamic: no calibration pass
        per_token = [v for k, v in sd.items() if k.endswith('.res_act_quantizer.activation_quantizer.quantizer._delta')]
        assert per_token and all(v.numel() == 128 for v in per_token)        # one range per token position
    assert any(k.endswith('weight_quantizer.range_estimator.quantizer._delta') for k in sd)
    if '--adaround' in flags:
        sda = torch.load(os.path.join(tmp_path, 'state_dict_adaround.pth'))
        assert any(k.endswith('.alpha') for k in sda), 'AdaRound state (alpha) must be in the checkpoint'
        assert 'adaround' in rep['timings_s']


@pytest.mark.gpu
def test_state_dict_round_trip_and_fast_inference(tmp_path, capsys):
    base = ['--qmethod', 'symmetric_uniform', '--qmethod-act', 'asymmetric_uniform', '--num-layers', '2',
            '--num-eval-batches', '2', '--num-est-batches', '2']
    a = V.main(base + ['--output-dir', str(tmp_path)])
    b = V.main(base + ['--load-state-dict', os.path.join(tmp_path, 'state_dict.pth')])
    assert b['load_state_dict']['unexpected'] == []
    assert abs(a['fidelity_vs_fp32']['logit_sqnr_db'] - b['fidelity_vs_fp32']['logit_sqnr_db']) < 1e-6
    d = V.main(base + ['--hip-graph'])                       # graph-replayed calibration + evaluation: same numbers
    assert abs(a['fidelity_vs_fp32']['logit_sqnr_db'] - d['fidelity_vs_fp32']['logit_sqnr_db']) < 1e-6
    # dynamic per-token inference (estimate + quantize at every site on every call) replays as ONE hipGraph, too
    dyn = base + ['--per-token', '--act-quant-method', 'current_minmax']
    e = V.main(dyn)
    g = V.main(dyn + ['--hip-graph'])
    assert e.get('dynamic') and g.get('dynamic')
    assert abs(e['fidelity_vs_fp32']['logit_sqnr_db'] - g['fidelity_vs_fp32']['logit_sqnr_db']) < 1e-6
    c = V.main(base + ['--load-state-dict', os.path.join(tmp_path, 'state_dict.pth'), '--fast-inference'])
    assert abs(a['fidelity_vs_fp32']['logit_sqnr_db'] - c['fidelity_vs_fp32']['logit_sqnr_db']) < 3.0
    from harness.bert import QResidualBlock, QSelfAttention
    from quantization import options
    QResidualBlock.fuse = QSelfAttention.fuse = False
    options.INT8_LINEAR = False


@pytest.mark.gpu
@pytest.mark.default_route
@pytest.mark.parametrize('flags', [
    ['--qmethod', 'symmetric_uniform', '--qmethod-act', 'asymmetric_uniform'],
    ['--qmethod', 'symmetric_uniform', '--qmethod-act', 'asymmetric_uniform', '--per-embd', '--act-quant-method', 'current_minmax'],
    ['--qmethod', 'symmetric_uniform', '--qmethod-act', 'asymmetric_uniform', '--n-bits', '4', '--n-bits-act', '8',
     '--quant-setup', 'FP_logits'],
    ['--qmethod', 'symmetric_uniform', '--per-channel'],
    # AdaRound collects its layer inputs / outputs through forward hooks on the optimised layer (adaround/utils.py): the
    # hooked layer must stay on the layered route while they are registered, whatever the switch says
    ['--qmethod', 'symmetric_uniform', '--qmethod-act', 'asymmetric_uniform', '--n-bits', '4', '--n-bits-act', '8',
     '--adaround', 'layers.0.output.dense,layers.1.intermediate.0', '--adaround-iters', '30', '--adaround-num-samples', '16'],
], ids=['w8a8', 'per-embd', 'w4a8-fp-logits', 'symmetric-acts-per-channel-weights', 'w4a8-adaround'])
def test_default_route_against_the_layered_route(flags, capsys):
    """The product default (options.INT8_LINEAR = 'auto') on configurations where the integer route applies fully (W8A8),
    partly (per-embedding sites keep per-axis ranges: only the softmax chain fuses; FP32 logits) or nowhere (symmetric
    activation grids): the evaluation must run, report the route, and stay within a dB of the layered route's fidelity."""
    from quantization import options
    assert options.INT8_LINEAR == 'auto'
    base = flags + ['--num-layers', '2', '--num-eval-batches', '2']
    auto = V.main(base)
    layered = V.main(base + ['--layered-inference'])
    assert auto['inference_route'] == "options.INT8_LINEAR = 'auto'" and layered['inference_route'] == 'layered (forced)'
    a, b = auto['fidelity_vs_fp32'], layered['fidelity_vs_fp32']
    assert a['samples'] == b['samples'] == 16
    assert abs(a['logit_sqnr_db'] - b['logit_sqnr_db']) < (3.0 if '--adaround' in flags else 1.5), (a, b)
    if '--adaround' in flags:
        assert 'adaround' in auto['timings_s'] and 'adaround' in layered['timings_s']
    if '--qmethod-act' not in flags:
        # symmetric activation quantizers: no integer plan, no fused tail eligible for an int8 index output -- but the fused
        # LayerNorm tail / softmax chain still apply (fp32 kernels), so equality is not expected, closeness is
        assert a['argmax_agreement'] >= b['argmax_agreement'] - 0.13
    assert options.INT8_LINEAR == 'auto'                    # the CLI restores the process-wide switches
