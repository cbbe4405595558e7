import sys, time, faulthandler
faulthandler.dump_traceback_later(240, exit=False)
sys.path.insert(0, '.')
order = sys.argv[1]
t0 = time.time()
def mark(what):
    print(f'{order}: {what} at {time.time() - t0:.2f} s', flush=True)
if order == 'torch-first':
    import torch
    mark('import torch')
    t = torch.ones(4, device='cuda')
    mark('torch cuda init')
import baseband_tasks_amd as bt
import numpy as np
a = bt.hip.DeviceArray.from_host(np.arange(4, dtype='float32'))
mark('library init + upload')
if order == 'lib-first':
    import torch
    mark('import torch')
    t = torch.ones(4, device='cuda')
    mark('torch cuda init')
print(float(t.sum()), a.to_host().sum())
