import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch
from test_models_gpu import interactions
from rectools_amd.dataset import Dataset
from rectools_amd.models import SASRecModel
ds = Dataset.construct(interactions())
kw = dict(n_factors=32, n_blocks=1, n_heads=2, session_max_len=3, lr=0.01, batch_size=4, dropout_rate=0.0, loss="softmax", seed=7)
def run(n):
    m = SASRecModel(epochs=n, **kw).fit(ds)
    return {k: v.detach().cpu().clone() for k,v in m.torch_model.state_dict().items()}, m.history
for n in (1,2,3):
    a,h1 = run(n); b,h2 = run(n)
    worst = max(float((a[k]-b[k]).abs().max()) for k in a)
    print("epochs",n,"max diff between two identical fits:",worst, [round(x['train_loss'],6) for x in h1], [round(x['train_loss'],6) for x in h2])
m = SASRecModel(epochs=3, **kw); m.fit_partial(ds, max_epochs=2); s2={k: v.detach().cpu().clone() for k,v in m.torch_model.state_dict().items()}
a2,_ = run(2)
print("fit_partial(2) vs fit(2):", max(float((a2[k]-s2[k]).abs().max()) for k in a2))
m.fit_partial(ds, max_epochs=1); s3={k: v.detach().cpu().clone() for k,v in m.torch_model.state_dict().items()}
a3,_ = run(3)
print("fit_partial(2+1) vs fit(3):", max(float((a3[k]-s3[k]).abs().max()) for k in a3), m.history)
