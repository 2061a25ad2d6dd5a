"""Synthetic-data RoBERTa harness on the drop-in quantization package: the BERT harness (harness/bert.py) with the two
things the reference's models/quantized_roberta.py changes.

* Position ids come from the input ids (reference models/quantized_roberta.py:26-41, 63-70): padding tokens keep
  position `padding_idx`, every other token counts up from `padding_idx + 1`.
* No pooler: the classification head (dense -> tanh -> out_proj on the first token) is quantized by the generic
  recursive rewriter (`quantize_model(org_model.classifier)`, :158).  Its tanh is a functional call, so the dense's
  output quantizer sees pre-tanh values and out_proj consumes the un-quantized tanh output; `quant_setup` has no
  effect on this head upstream (the BERT constructor's choice is overwritten, :155-158) and is not offered here.

The encoder layers, their 13 quantizer sites each and every fused fast path are the BERT ones (QLayer).
"""
import torch
from torch import nn

from quantization.autoquant_utils import quantize_model
from quantization.base_quantized_model import QuantizedModel

from harness.bert import QEmbeddings, QLayer


class QRobertaEmbeddings(QEmbeddings):
    def __init__(self, hf, **qp):
        super().__init__(hf, **qp)
        self.padding_idx = hf.padding_idx

    def position_ids(self, input_ids):
        keep = input_ids.ne(self.padding_idx).int()
        return (torch.cumsum(keep, dim=1).type_as(keep) * keep).long() + self.padding_idx


class QRobertaForSequenceClassification(QuantizedModel):
    def __init__(self, hf, **qp):
        super().__init__()
        self.embeddings = QRobertaEmbeddings(hf.roberta.embeddings, **qp)
        self.layers = nn.ModuleList([QLayer(l, **qp) for l in hf.roberta.encoder.layer])
        self.classifier = quantize_model(hf.classifier, **qp)      # HF head: features[:, 0] -> dense -> tanh -> out_proj

    def forward(self, input_ids, attention_mask=None):
        if attention_mask is not None:
            mask = (1.0 - attention_mask[:, None, None, :].float()) * -10000.0
        else:
            from harness.bert import constant
            mask = constant('mask', input_ids.shape[0], input_ids.shape[1], input_ids.device)
        h = self.embeddings(input_ids)
        for layer in self.layers:
            h = layer(h, mask)
        return self.classifier(h)


def build_roberta(seed=1000, num_labels=2, num_layers=None, **qp):
    """HF RoBERTa-base architecture with parameters from the build-independent numpy stream (harness/weights.py),
    wrapped with quantizers."""
    from transformers import RobertaConfig, RobertaForSequenceClassification
    from harness.weights import fill_from_numpy_stream
    torch.manual_seed(seed)
    cfg = RobertaConfig(num_labels=num_labels, vocab_size=50265, max_position_embeddings=514, type_vocab_size=1,
                        pad_token_id=1)
    if num_layers is not None:
        cfg.num_hidden_layers = num_layers
    hf = fill_from_numpy_stream(RobertaForSequenceClassification(cfg).eval(), seed)
    return QRobertaForSequenceClassification(hf, **qp), hf
