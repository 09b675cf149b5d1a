"""Independent Python statements used to pin ganon_amd/host/robin_order.hpp (the reference's robin_hood map order):

  murmur64a            Austin Appleby's MurmurHash64A as published (MurmurHash2.cpp) -- libstdc++'s std::hash<std::string> is
                       this function with seed 0xc70f6907 (the tests also ask the real libstdc++ through a C++ helper)
  rh_hash_bytes        robin_hood::hash_bytes = the same function, seed 0xe17a1465, WITHOUT its last multiply/shift
  RobinTable           robin_hood::detail::Table's slot bookkeeping (3.11 series), written over a plain list of
                       (info, key, hash) cells instead of the C++ class's parallel arrays
"""
M64 = (1 << 64) - 1
MUL = 0xc6a4a7935bd1e995
R = 47


def _body(data: bytes, seed: int) -> int:
    n = len(data)
    h = (seed ^ (n * MUL)) & M64
    for i in range(0, n - n % 8, 8):
        k = int.from_bytes(data[i:i + 8], "little")
        k = (k * MUL) & M64
        k ^= k >> R
        k = (k * MUL) & M64
        h ^= k
        h = (h * MUL) & M64
    tail = data[n - n % 8:]
    if tail:
        h ^= int.from_bytes(tail, "little")
        h = (h * MUL) & M64
    return h


def murmur64a(data: bytes, seed: int) -> int:
    h = _body(data, seed)
    h ^= h >> R
    h = (h * MUL) & M64
    h ^= h >> R
    return h


def std_hash(data: bytes) -> int:
    return murmur64a(data, 0xc70f6907)


def rh_hash_bytes(data: bytes) -> int:
    h = _body(data, 0xe17a1465)
    return h ^ (h >> R)


def pair_hash(a: bytes, b: bytes) -> int:
    return (std_hash(a) ^ (std_hash(b) << 1)) & M64


class RobinTable:
    """cells[i] = [info, key, hash]; info 0 = empty.  80 % load factor, 5 info bits to start with."""

    def __init__(self):
        self.cells = None
        self.count = 0
        self.mask = 0
        self.limit = 0
        self.inc = 32
        self.shift = 0
        self.mult = 0xc4ceb9fe1a85ec53

    @staticmethod
    def _limit(n):
        return n * 80 // 100

    def _span(self, n):
        return n + min(self._limit(n), 0xFF)

    def _alloc(self, n):
        self.count, self.mask, self.limit = 0, n - 1, self._limit(n)
        self.cells = [[0, None, 0] for _ in range(self._span(n))] + [[1, None, 0]]   # sentinel
        self.inc, self.shift = 32, 0

    def _home(self, h):
        h = (h * self.mult) & M64
        h ^= h >> 33
        return (h >> 5) & self.mask, self.inc + ((h & 31) >> self.shift)

    def _more_info_bits(self):
        if self.inc <= 2:
            return False
        self.inc >>= 1
        self.shift += 1
        for c in self.cells[:-1]:
            c[0] = (c[0] >> 1) & 0x7F
        self.cells[-1][0] = 1
        self.limit = self._limit(self.mask + 1)
        return True

    def _grow(self):
        if self.cells is None:
            self._alloc(8)
            return
        n = self.mask + 1
        if self.count < self._limit(n) and self._more_info_bits():
            return
        old = [c for c in self.cells[:-1] if c[0]]
        if self.count * 2 < self._limit(n):
            self.mult = (self.mult + 0xc4ceb9fe1a85ec54) & M64
            self._alloc(n)
        else:
            self._alloc(2 * n)
        for _, key, h in old:
            self._place(key, h, moving=True)

    def _place(self, key, h, moving):
        if moving and self.limit == 0 and not self._more_info_bits():
            raise OverflowError
        idx, info = self._home(h)
        if moving:
            while info <= self.cells[idx][0]:
                idx, info = idx + 1, info + self.inc
        at, at_info = idx, info & (0xFF if moving else 0xFFFFFFFF)
        if at_info + self.inc > 0xFF:
            self.limit = 0
        while self.cells[idx][0]:
            idx += 1
        # shift cells [at, idx) up by one; their infos grow by one step
        for j in range(idx, at, -1):
            self.cells[j][1], self.cells[j][2] = self.cells[j - 1][1], self.cells[j - 1][2]
        for j in range(idx, at, -1):
            self.cells[j][0] = (self.cells[j - 1][0] + self.inc) & 0xFF
            if self.cells[j][0] + self.inc > 0xFF:
                self.limit = 0
        self.cells[at] = [at_info & 0xFF, key, h]
        self.count += 1

    def insert(self, key, h) -> bool:
        for _ in range(256):
            if self.cells is None:
                self._grow()
                continue
            idx, info = self._home(h)
            while info < self.cells[idx][0]:
                idx, info = idx + 1, info + self.inc
            while info == self.cells[idx][0]:
                if self.cells[idx][1] == key:
                    return False
                idx, info = idx + 1, info + self.inc
            if self.count >= self.limit:
                self._grow()
                continue
            # (same walk as above leads to the same cell)
            at, at_info = idx, info
            if at_info + self.inc > 0xFF:
                self.limit = 0
            while self.cells[idx][0]:
                idx += 1
            for j in range(idx, at, -1):
                self.cells[j][1], self.cells[j][2] = self.cells[j - 1][1], self.cells[j - 1][2]
            for j in range(idx, at, -1):
                self.cells[j][0] = (self.cells[j - 1][0] + self.inc) & 0xFF
                if self.cells[j][0] + self.inc > 0xFF:
                    self.limit = 0
            self.cells[at] = [at_info & 0xFF, key, h]
            self.count += 1
            return True
        raise OverflowError

    def order(self):
        return [] if self.cells is None else [c[1] for c in self.cells[:-1] if c[0]]
