"""The BERT harness lives in the package (transformer-quantization_amd/harness/bert.py) since the
validate_quantized CLI uses it too; tests keep importing it from here."""
from harness.bert import *          # noqa: F401,F403
from harness.bert import QBertForSequenceClassification, QResidualBlock, QSelfAttention, build_bert_base, quantizer_census  # noqa: F401
