import sys, os
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
os.environ['TQ_FFN_DBG']='1'
import pytest
sys.exit(pytest.main(['/root/repo/tests/test_linear_i8.py','-q','-m','gpu','-k','ffn_i8 and 1024 and all and False','-s','-x']))
