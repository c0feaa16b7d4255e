import sys, os
sys.path.insert(0, os.getcwd())
from deepdish_amd.multipipe import MultiStreamPipeline
for model in ('synthetic-ssd_mobilenet_v1-uint8', 'synthetic-ssd_mobilenet_v1'):
    mp = MultiStreamPipeline(384, model=model)
    print(model, 'detector GB', mp.det.activation_bytes() / 1e9, 'encoder GB (shared)', mp.enc.activation_bytes() / 1e9)
    del mp
os.environ['DD_NET_SHARED'] = '0'
mp = MultiStreamPipeline(384)
print('encoder GB (one buffer per tensor)', mp.enc.activation_bytes() / 1e9)
