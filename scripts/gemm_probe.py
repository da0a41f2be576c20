import torch, time
dev="cuda"
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
M=200000
for K in (71,72,80,96,128):
    x=torch.randn(M,K,device=dev); w=torch.randn(256,K,device=dev); b=torch.randn(256,device=dev)
    print("fp32 linear M=%d K=%d N=256: %.1f us" % (M,K,t(lambda: torch.nn.functional.linear(x,w,b))))
x=torch.randn(M,256,device=dev); w=torch.randn(256,256,device=dev); b=torch.randn(256,device=dev)
print("fp32 linear 256->256: %.1f us" % t(lambda: torch.nn.functional.linear(x,w,b)))
print("relu 200k x 256: %.1f us" % t(lambda: torch.relu(x)))
w2=torch.randn(21,256,device=dev); b2=torch.randn(21,device=dev)
print("fp32 linear 256->21: %.1f us" % t(lambda: torch.nn.functional.linear(x,w2,b2)))
w3=torch.randn(43,256,device=dev); b3=torch.randn(43,device=dev)
print("fp32 linear 256->43 (heads fused): %.1f us" % t(lambda: torch.nn.functional.linear(x,w3,b3)))
y=torch.randn(M,21,device=dev)
print("softmax 200k x 21: %.1f us" % t(lambda: torch.softmax(y,-1)))
xb=x.bfloat16(); wb=w.bfloat16(); bb=b.bfloat16()
print("bf16 linear 256->256: %.1f us" % t(lambda: torch.nn.functional.linear(xb,wb,bb)))
# fused bias + relu epilogue (hipBLASLt) vs linear followed by relu
x1=torch.randn(M,71,device=dev); w1=torch.randn(256,71,device=dev); b1=torch.randn(256,device=dev)
try:
    f=lambda: torch._addmm_activation(b1, x1, w1.t(), use_gelu=False)
    ref=torch.relu(torch.nn.functional.linear(x1,w1,b1))
    got=f()
    print("addmm_activation(relu) 71->256: %.1f us, max abs diff vs linear+relu %.3g" % (t(f), (got-ref).abs().max().item()))
    f2=lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False)
    print("addmm_activation(relu) 256->256: %.1f us" % t(f2))
except Exception as e:
    print("addmm_activation failed:", repr(e)[:200])
print("linear+relu 71->256: %.1f us" % t(lambda: torch.relu(torch.nn.functional.linear(x1,w1,b1))))
print("relu_ inplace 200k x 256: %.1f us" % t(lambda: torch.relu_(x)))
