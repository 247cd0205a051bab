import os, ctypes, torch, collections
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools/probes/libhwid.so"))
out = torch.zeros(256 * 8, dtype=torch.int32, device="cuda")
rc = lib.hwid_launch(ctypes.c_void_p(out.data_ptr()), 256, 151552, None)
torch.cuda.synchronize()
v = out.cpu().view(256, 8)
for b in (0, 1, 100, 255):
    print(b, [(int(x) & 0xF, (int(x) >> 4) & 3, (int(x) >> 8) & 0xF, (int(x) >> 12) & 1, (int(x) >> 13) & 7) for x in v[b]], "(wave, simd, cu, sh, se)")
c = collections.Counter(tuple(sorted(((int(x) >> 4) & 3) for x in row)) for row in v)
print("simd multiset per workgroup:", c)
