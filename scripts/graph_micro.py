import time, torch
dev='cuda'
x=torch.zeros(1024,device=dev)
def body(n):
    for _ in range(n): x.add_(1)
for n in (100,1000,4000):
    s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): body(10)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): body(n)
    g.replay(); torch.cuda.synchronize()
    t0=time.perf_counter(); 
    for _ in range(5): g.replay()
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    torch.cuda.synchronize()
    t3=time.perf_counter(); 
    for _ in range(5): body(n)
    t4=time.perf_counter(); torch.cuda.synchronize(); t5=time.perf_counter()
    print("n=%d graph: host %.2f ms total %.2f ms (%.2f us/node) | eager: host %.2f total %.2f (%.2f us/op)"%(n,(t1-t0)/5*1e3,(t2-t0)/5*1e3,(t2-t0)/5/n*1e6,(t4-t3)/5*1e3,(t5-t3)/5*1e3,(t5-t3)/5/n*1e6))
