"""Diagnostic: Regev enc+dec steps run back to back on one stream (no events: 0.289 ms/step) vs software-pipelined on two (slower: 0.313)
(decode of batch k on stream B while the GEMM of batch k+1 runs on stream A; ciphertexts double
buffered)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crypto_amd.device import Context, X_BITS
n, m, q, B = 256, 6168, 16777213, 65536
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(sA):
    cA = Context(0, sA)
with torch.cuda.stream(sB):
    cB = Context(0, sB)
pk = cA.sample_uniform(1, 1, 0, (n + 1) * m, q).view(n + 1, m)
s = cA.sample_uniform(1, 2, 0, n, q)
w = ((q - (s.to(torch.int64) & 0xFFFFFFFF)) % q).to(torch.int32)
key = cA.key_upload(pk, q)
xb = cA.sample_bits(1, 3, 0, B, m)
cipher = [cA.empty_u32(B, n + 1), cA.empty_u32(B, n + 1)]
bits = [cA.empty_u8(B), cA.empty_u8(B)]
torch.cuda.synchronize()

def sequential(steps):
    with torch.cuda.stream(sA):
        for k in range(steps):
            cA.matmul(key, xb, X_BITS, B, out=cipher[0])
            cA.dot_decode(w, q, cipher[0], out=bits[0])

def pipelined(steps):
    done_enc = [torch.cuda.Event() for _ in range(2)]
    done_dec = [torch.cuda.Event() for _ in range(2)]
    for k in range(steps):
        i = k & 1
        with torch.cuda.stream(sA):
            if k >= 2:
                sA.wait_event(done_dec[i])          # buffer i is free again
            cA.matmul(key, xb, X_BITS, B, out=cipher[i])
            done_enc[i].record(sA)
        with torch.cuda.stream(sB):
            sB.wait_event(done_enc[i])
            cB.dot_decode(w, q, cipher[i], out=bits[i])
            done_dec[i].record(sB)

for name, fn in (("sequential", sequential), ("pipelined", pipelined), ("sequential", sequential), ("pipelined", pipelined)):
    fn(200); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(400); torch.cuda.synchronize()
    print("%-10s %.4f ms/step" % (name, (time.perf_counter() - t0) / 400 * 1e3), flush=True)
