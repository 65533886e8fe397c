import math, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from adapter4rec_amd import _lib as L
dev = 'cuda:0'
for S, nh, dh in [(40, 1, 128), (40, 1, 64), (100, 2, 128)]:
    n = 2
    H = nh * dh
    Mp = ((n * S + 255) // 256) * 256
    g = torch.Generator().manual_seed(S)
    qkv = torch.randn(Mp, 3 * H, generator=g).to(dev)
    scale = 1 / math.sqrt(dh)
    for mode in ('plain', 'mask', 'causal'):
        km = torch.ones(n, S, device=dev)
        if mode != 'plain':
            km[1, :S // 2] = 0
        out = torch.zeros(Mp, H, device=dev); lse = torch.zeros(n * nh * S, device=dev)
        L.attn_long_fwd(qkv, out, lse, n, S, nh, dh, 0, H, 2 * H, scale, key_mask=None if mode == 'plain' else km, causal=mode == 'causal')
        xr = qkv[:n * S].double().clone().requires_grad_(True)
        x = xr.view(n, S, 3, nh, dh)
        q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
        sc = q @ k.transpose(-1, -2) * scale
        allowed = (km != 0)[:, None, None, :].expand(n, 1, S, S)
        if mode == 'causal':
            allowed = torch.tril(allowed)
        sc = sc + torch.where(allowed, 0.0, -1e9).double()
        ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(n * S, H)
        rows_ok = allowed.any(-1).reshape(n, S).reshape(-1) if mode != 'plain' else torch.ones(n * S, dtype=torch.bool, device=dev)
        err = (out[:n * S].double() - ref.detach()).abs()[rows_ok]
        dout = torch.randn(Mp, H, generator=g).to(dev)
        dout[n * S:] = 0
        dout[:n * S] *= rows_ok[:, None]
        dqkv = torch.zeros_like(qkv); ws = torch.zeros_like(lse)
        L.attn_long_bwd(qkv, out, dout, dqkv, lse, ws, n, S, nh, dh, 0, H, 2 * H, scale, key_mask=None if mode == 'plain' else km, causal=mode == 'causal')
        ref.backward(dout[:n * S].double())
        e = (dqkv[:n * S].double() - xr.grad).abs()
        print(S, nh, dh, mode, 'fwd', f'{float(err.max()):.1e}', 'dq', f'{float(e[:, :H].max()):.1e}', 'dk', f'{float(e[:, H:2 * H].max()):.1e}', 'dv', f'{float(e[:, 2 * H:].max()):.1e}',
              'dk by 16 cols', [round(float(e[:, H + c:H + c + 16].max()), 3) for c in range(0, dh, 16)])
