import torch, time, sys
sys.path.insert(0,'/root/repo')
from stylerenderer_amd.op.conv import conv2d_mfma
dev='cuda'
def run(b,c,n,res,k=3,stride=1,pad=1,tr=False, iters=5):
    x=torch.randn(b,c,res,res,device=dev); wt=torch.randn(k*k,c,n,device=dev)
    isc=torch.randn(b,c,device=dev); osc=torch.randn(b,n,device=dev)
    for _ in range(2): y=conv2d_mfma(x,wt,isc,osc,None,k,stride,pad,tr)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(iters): y=conv2d_mfma(x,wt,isc,osc,None,k,stride,pad,tr)
    torch.cuda.synchronize(); dt=(time.time()-t)/iters
    fl = 2*b*res*res*c*n*k*k if not tr else 2*b*res*res*c*n*9
    if stride==2 and not tr: fl/=4
    print(f"B{b} C{c} N{n} res{res} k{k} s{stride} tr{tr}: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TFLOP/s", flush=True)
run(16,128,128,256); run(16,256,256,128); run(16,512,512,64); run(16,512,512,32); run(16,512,512,16); run(16,512,512,8); run(16,512,512,4)
run(16,256,128,128,3,2,0,True); run(16,512,256,64,3,2,0,True); run(16,512,512,32,3,2,0,True)
run(16,128,256,257,3,2,0,False)

from stylerenderer_amd.op.conv import conv2d_wgrad_mfma
def runw(b,c,n,res,k=3,stride=1,pad=1,tr=False, iters=5):
    x=torch.randn(b,c,res,res,device=dev)
    oh = 2*res+1 if tr else (res+2*pad-k)//stride+1
    gy=torch.randn(b,n,oh,oh,device=dev)
    isc=torch.randn(b,c,device=dev); osc=torch.randn(b,n,device=dev)
    for _ in range(2): y=conv2d_wgrad_mfma(x,gy,isc,osc,k,stride,pad,tr)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(iters): y=conv2d_wgrad_mfma(x,gy,isc,osc,k,stride,pad,tr)
    torch.cuda.synchronize(); dt=(time.time()-t)/iters
    fl = 2*b*res*res*c*n*9 if tr else 2*b*oh*oh*c*n*k*k
    print(f"WGRAD B{b} C{c} N{n} res{res} k{k} s{stride} tr{tr}: {dt*1e3:.3f} ms  {fl/dt/1e12:.1f} TFLOP/s", flush=True)
runw(16,128,128,256); runw(16,256,256,128); runw(16,512,512,64); runw(16,512,512,32); runw(16,512,512,16); runw(16,512,512,8); runw(16,512,512,4)
runw(16,256,128,128,3,2,0,True); runw(16,512,256,64,3,2,0,True); runw(16,512,512,32,3,2,0,True)
