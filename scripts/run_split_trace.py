import sys, torch, numpy as np
sys.path.insert(0, ".")
import rnnt_speech_recognition_amd as pkg
dev = torch.device("cuda:0")
B,T,U,V = 32,600,150,28
g = torch.Generator().manual_seed(1)
acts = torch.randn(B,T,U,V, generator=g).to(dev)
labels = torch.randint(1,V,(B,U-1),generator=g,dtype=torch.int32).to(dev)
il = torch.full((B,),T,dtype=torch.int32,device=dev); ll = torch.full((B,),U-1,dtype=torch.int32,device=dev)
for _ in range(3):
    c, gr = pkg.rnnt_loss_and_grad(acts, labels, il, ll)
torch.cuda.synchronize()
print(float(c.sum()))
