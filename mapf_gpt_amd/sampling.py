"""Host restatement of the library's action sampler (csrc/gpt.hip: uniform01 + sample_kernel), used
by tests to verify the device sampler and by callers that want reproducible draws off-device.

The reference samples with torch.multinomial (model.py:257) whose stream is device-specific; this
package keys a counter-based RNG by (seed, step, row) instead, so a draw does not depend on batch
chunking or on how instances are sharded over GPUs."""
import numpy as np

_M = (1 << 64) - 1


def uniform01(seed, step, rows):
    """rows: int array of global row ids -> float32 uniforms in [0,1) with 24 random bits."""
    rows = np.asarray(rows, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64((int(seed) + 0x9E3779B97F4A7C15 * (int(step) + 1)) & _M)
        z = z ^ (rows * np.uint64(0xD1342543DE82EF95))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def sample(logits, seed, step, row0=0, do_sample=True):
    """logits float32 [rows, >=5] -> (actions int32 [rows], margin float32 [rows]); margin = distance of the
    uniform draw to the nearest CDF edge (tests skip rows whose margin is below float round-off)."""
    l = np.asarray(logits, dtype=np.float32)[:, :5]
    best = np.argmax(l, axis=1).astype(np.int32)
    if not do_sample:
        return best, np.full(len(l), np.inf, dtype=np.float32)
    mx = l.max(axis=1, keepdims=True)
    e = np.exp((l - mx).astype(np.float32)).astype(np.float32)
    tot = np.zeros(len(l), dtype=np.float32)
    for i in range(5):
        tot = (tot + e[:, i]).astype(np.float32)
    u = (uniform01(seed, step, np.arange(len(l)) + row0) * tot).astype(np.float32)
    c = np.zeros(len(l), dtype=np.float32)
    act = np.full(len(l), 4, dtype=np.int32)
    done = np.zeros(len(l), dtype=bool)
    margin = np.full(len(l), np.inf, dtype=np.float32)
    for i in range(5):
        c = (c + e[:, i]).astype(np.float32)
        margin = np.minimum(margin, np.abs(u - c) / tot)
        hit = (~done) & (u < c)
        act[hit] = i
        done |= hit
    return act, margin
