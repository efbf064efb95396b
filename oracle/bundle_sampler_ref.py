"""TEST INFRASTRUCTURE — CPU restatement of the device bundle sampler (paccmann_polymer_amd/csrc/ppk_bundles.hip).

The reference samples a node's 2-hop bundle with numpy's global RNG
(/root/reference/paccmann_polymer/topologically_regularized_models/experiments/chem_reactions/utils.py:155-196); the
device kernel keeps the reference's statements and replaces the random stream by one Philox4x32-10 stream per root.
`PhiloxRng` below is that stream behind the two numpy calls the reference makes (`permutation`, `choice`), so the HOST
restatement of the reference's statements (data_prep.sample_bundle / relabel_bundle, themselves checked against
networkx in tests/test_data_prep.py) can be run on exactly the device's draws: outputs must agree bit for bit.
Only tests/ may import this module."""
M32 = 0xFFFFFFFF


def philox4x32_10(c, k0, k1):
    c = list(c)
    for _ in range(10):
        p0 = 0xD2511F53 * c[0]
        p1 = 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c[3] ^ k1) & M32, p0 & M32]
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return c


class PhiloxRng:
    """The draws of root `root` under `seed`: draw i = (first Philox word * n) >> 32, counter = (i, 0, root lo, root hi)."""

    def __init__(self, seed, root, max_close):
        self.seed, self.root, self.max_close, self.n = int(seed), int(root), int(max_close), 0

    def _draw(self, n):
        c = philox4x32_10([self.n, 0, self.root & M32, (self.root >> 32) & M32], self.seed & M32, (self.seed >> 32) & M32)
        self.n += 1
        return (c[0] * n) >> 32

    def permutation(self, items):
        """Only the first max_close entries are ever read (utils.py:161 slices them off): draw i takes the d-th
        position that is still free; the rest follows in list order."""
        items = list(items)
        free = list(range(len(items)))
        head = []
        for i in range(min(self.max_close, len(items))):
            head.append(free.pop(self._draw(len(items) - i)))
        return [items[p] for p in head] + [items[p] for p in free]

    def choice(self, items):
        items = list(items)
        return items[self._draw(len(items))]
