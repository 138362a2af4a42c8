import ctypes, torch, collections
L = ctypes.CDLL("tests/ldspoison/liblds_poison.so")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
out = torch.zeros(1024, dtype=torch.int32, device="cuda")
for idx in (0, 1000, 30000, 40000):
    print("poison rc", L.lds_poison(st)); torch.cuda.synchronize()
    print("peek rc", L.lds_peek(idx, 1024, ctypes.c_void_p(out.data_ptr()), st)); torch.cuda.synchronize()
    c = collections.Counter(("%08x" % (v & 0xffffffff)) for v in out.cpu().tolist())
    print(idx, c.most_common(4))
