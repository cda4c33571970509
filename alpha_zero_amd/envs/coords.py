"""Coordinate conversions of the board envs (reference: alpha_zero/envs/coords.py:45-91).
flat index = row * N + col, N*N = pass; SGF 'aa' = top-left, column letter first; GTP skips 'I', rows count from the bottom."""

_SGF = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"
_GTP = "ABCDEFGHJKLMNOPQRSTUVWXYZ"


class CoordsConvertor:
    def __init__(self, board_size) -> None:
        self.board_size = board_size

    def from_flat(self, flat):
        n = self.board_size
        return None if flat == n * n else divmod(flat, n)

    def to_flat(self, coord):
        n = self.board_size
        return n * n if coord is None else n * coord[0] + coord[1]

    def from_sgf(self, sgfc):
        if sgfc is None or sgfc == "" or (self.board_size <= 19 and sgfc == "tt"):
            return None
        return _SGF.index(sgfc[1]), _SGF.index(sgfc[0])

    def to_sgf(self, coord):
        return "" if coord is None else _SGF[coord[1]] + _SGF[coord[0]]

    def from_gtp(self, gtpc):
        gtpc = gtpc.upper()
        if gtpc == "PASS":
            return None
        return self.board_size - int(gtpc[1:]), _GTP.index(gtpc[0])

    def to_gtp(self, coord):
        if coord is None:
            return "pass"
        return "{}{}".format(_GTP[coord[1]], self.board_size - coord[0])
