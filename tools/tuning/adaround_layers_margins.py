import sys
sys.path.insert(0,'/root/repo/transformer-quantization_amd'); sys.path.insert(0,'/root/repo')
import copy, json, numpy as np, torch
from tests.conftest import load_golden
from tests.test_adaround_layers import _model
from quantization.adaround import apply_adaround_to_layer
from quantization.adaround.config import DEFAULT_ADAROUND_CONFIG
z,_=load_golden('adaround_layers'); meta=json.loads(str(z['meta']))
ids=torch.from_numpy(z['ids']).cuda()
for c in meta['cases']:
    k,lname=c['k'],c['layer']
    model=_model(meta,ids,'cuda'); layer=getattr(model,lname)
    cfg=copy.deepcopy(DEFAULT_ADAROUND_CONFIG); cfg.iters,cfg.lr=c['iters'],c['lr']
    model.full_precision(); layer.quantized_weights(); torch.manual_seed(c['seed'])
    res=apply_adaround_to_layer(model,layer,ids,batch_size=c['bs'],act_quant=False,adaround_config=cfg)
    wq=layer.weight_quantizer.quantizer; alpha=wq.alpha.detach()
    with torch.no_grad(): idx=wq.to_integer_forward(layer.weight)
    got=np.array([res.loss_soft_before,res.loss_hard_before,res.loss_soft_after,res.loss_hard_after]); ref=z[f'l{k}_losses']
    print(lname,'loss rel dev',np.abs(got[1:]/ref[1:]-1), 'soft0',got[0],ref[0])
    lr=c['lr']
    if lname=='emb':
        rows=torch.from_numpy(z[f'l{k}_rows']); a=alpha[rows.cuda()].cpu(); ar=torch.from_numpy(z[f'l{k}_alpha_rows'])
        dev=(a-ar).abs()
        print(' touched rows: frac<=3lr',float((dev[:48]<=3*lr).float().mean()),'max/lr',float(dev[:48].max())/lr,'untouched max',float(dev[48:].max()))
        print(' ups diff',int((alpha>=0).sum())-int(z[f'l{k}_ups']),'bad rows',int((idx.sum(1).to(torch.int64).cpu().numpy()!=z[f'l{k}_hard_rowsum']).sum()))
    else:
        dev=(alpha.cpu()-torch.from_numpy(z[f'l{k}_alpha'])).abs()
        print(' frac<=3lr',float((dev<=3*lr).float().mean()),'max/lr',float(dev.max())/lr,'flips',int((idx.cpu()!=torch.from_numpy(z[f'l{k}_hard_idx'])).sum()))
