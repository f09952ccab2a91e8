"""CPU study for the NEXT lever of the wide convolutions: they run at the power-limited matrix throughput, so what moves
them is fewer FLOPs.  Winograd F(2x2, 3x3) needs 16 instead of 36 multiplications per output pair of rows - 2.25 x fewer MFMA FLOPs - at the price of
transformed operands that must be rounded to 16 bits again before the matrix cores see them.  This script prices that rounding: a 320-channel 3x3
convolution (the ConvGRU gates' shape) in fp64, with fp16 operands (what conv3x3_big_kernel computes), and as Winograd with the transformed
inputs / filters rounded to fp16 or bf16, products accumulated exactly (fp32 accumulation is not the limit here).
    python tools/winograd_error_study.py      (CPU, seconds)"""
import torch, math
torch.manual_seed(0)
# F(2x2, 3x3) Winograd: Y = A^T [ (G g G^T) .* (B^T d B) ] A
Bt = torch.tensor([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], dtype=torch.float64)
G  = torch.tensor([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], dtype=torch.float64)
At = torch.tensor([[1,1,1,0],[0,1,-1,-1]], dtype=torch.float64)
def wino(x, w, rnd):
    # x [Cin,H,W] (H,W even+2 halo), w [Cout,Cin,3,3]; rnd: rounding applied to transformed operands (what the MFMA would see)
    Cin, H, W = x.shape; Cout = w.shape[0]
    U = torch.einsum("ij,ocjk,lk->ocil", G, w.double(), G)            # [Cout,Cin,4,4]
    U = rnd(U)
    th, tw = (H - 2) // 2, (W - 2) // 2
    tiles = x.double().unfold(1, 4, 2).unfold(2, 4, 2)                 # [Cin,th,tw,4,4]
    V = torch.einsum("ij,cabjk,lk->cabil", Bt, tiles, Bt)
    V = rnd(V)
    M = torch.einsum("ocil,cabil->oabil", U.float().double(), V.float().double())   # fp32-accumulate modelled as exact (fp64) sum of fp16 products
    Y = torch.einsum("ij,oabjk,lk->oabil", At, M, At)                 # [Cout,th,tw,2,2]
    return Y.permute(0,1,3,2,4).reshape(Cout, th*2, tw*2)
h = lambda t: t.half().double()
ident = lambda t: t
Cin, Cout, H, W = 320, 64, 18, 18
for name, xs in (("unit normal activations", 1.0), ("post-ReLU-like (abs normal)", None)):
    x = torch.randn(Cin, H, W) if xs else torch.randn(Cin, H, W).abs()
    w = torch.randn(Cout, Cin, 3, 3) * (1.0 / math.sqrt(Cin * 9))
    ref = torch.nn.functional.conv2d(x.double()[None], w.double())[0]
    direct16 = torch.nn.functional.conv2d(h(x)[None], h(w))[0]        # fp16 operands, exact accumulate
    w16 = wino(h(x), h(w), h)                                          # inputs already fp16 (as stored), transformed operands rounded to fp16
    wbf = wino(h(x), h(w), lambda t: t.bfloat16().double())
    e = lambda y: ((y - ref).abs().mean() / ref.abs().mean()).item()
    print("%-30s rel mean err: direct fp16 operands %.2e | Winograd, transformed operands fp16 %.2e | ... bf16 %.2e | output scale %.2f" % (name, e(direct16), e(w16), e(wbf), ref.abs().mean()))
# exactness check of the transform
x = torch.randn(8, 10, 10); w = torch.randn(4, 8, 3, 3)
print("transform exact:", (wino(x, w, ident) - torch.nn.functional.conv2d(x.double()[None], w.double())[0]).abs().max().item())
