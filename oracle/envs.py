"""ctypes wrappers over oracle/rules.c exposing the attribute surface that the
reference search and actor read from an env (mcts_v2.py:356-448,
pipeline.py:300-380).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
MAXN, MAXP, NUM_STACK = 19, 361, 8


class _EnvStruct(ctypes.Structure):
    _fields_ = [
        ("kind", ctypes.c_int32), ("n", ctypes.c_int32), ("num_to_win", ctypes.c_int32), ("max_steps", ctypes.c_int32),
        ("komi", ctypes.c_double),
        ("board", ctypes.c_int8 * MAXP),
        ("legal", ctypes.c_int8 * (MAXP + 1)),
        ("hist", (ctypes.c_int8 * MAXP) * NUM_STACK),
        ("ko", ctypes.c_int16), ("caps", ctypes.c_int16 * 2), ("steps", ctypes.c_int16),
        ("last_move", ctypes.c_int16), ("prev_move", ctypes.c_int16), ("hist_last", ctypes.c_int16),
        ("to_play", ctypes.c_int8), ("last_player", ctypes.c_int8), ("winner", ctypes.c_int8), ("done", ctypes.c_int8),
    ]


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(so)
        assert L.oracle_env_sizeof() == ctypes.sizeof(_EnvStruct), "oracle struct layout mismatch"
        P = ctypes.POINTER(_EnvStruct)
        L.oracle_env_init.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int]
        L.oracle_env_reset.argtypes = [P]
        L.oracle_env_step.argtypes = [P, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
        L.oracle_env_observation.argtypes = [P, ctypes.c_void_p]
        L.oracle_env_record.argtypes = [P, ctypes.c_double, ctypes.c_void_p]
        L.oracle_go_area_score.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.oracle_go_score.argtypes = [P]
        L.oracle_go_score.restype = ctypes.c_double
        L.oracle_env_replay.argtypes = [P, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64),
                                        ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64)]
        _LIB = L
    return _LIB


class OracleEnv:
    """Common part of the two oracle envs (mirrors BoardGameEnv, base.py:26-112)."""

    kind = 0
    has_pass_move = False
    has_resign_move = False
    black_player, white_player = 1, 2
    legal_dtype = np.int8

    def __init__(self, board_size, komi=7.5, max_steps=0, num_to_win=5, num_stack=8):
        assert num_stack == 8
        self._s = _EnvStruct()
        self.board_size = board_size
        self.num_stack = num_stack
        self.action_dim = board_size * board_size + (1 if self.has_pass_move else 0)
        self.pass_move = self.action_dim - 1 if self.has_pass_move else None
        self.resign_move = -1 if self.has_resign_move else None
        lib().oracle_env_init(ctypes.byref(self._s), self.kind, board_size, komi, max_steps, num_to_win)
        self.history = []

    # ---- reference attribute surface -------------------------------------------------
    @property
    def to_play(self):
        return int(self._s.to_play)

    @property
    def opponent_player(self):
        return self.white_player if self.to_play == self.black_player else self.black_player

    @property
    def last_player(self):
        return int(self._s.last_player) or None

    @property
    def last_move(self):
        return None if self._s.last_move == -2 else int(self._s.last_move)

    @property
    def winner(self):
        return int(self._s.winner) or None

    @property
    def steps(self):
        return int(self._s.steps)

    @property
    def board(self):
        n = self.board_size
        return np.frombuffer(self._s.board, dtype=np.int8, count=n * n).reshape(n, n).copy()

    @property
    def legal_actions(self):
        # dtype matters downstream: Go masks are int64 (go_engine.py:441 concatenates with a Python
        # list), Gomoku/terminal masks are int8 (base.py:72) -> decides float64 vs float32 search_pi.
        a = np.frombuffer(self._s.legal, dtype=np.int8, count=self.action_dim)
        return a.astype(np.int8 if self._s.done else self.legal_dtype)

    def is_game_over(self):
        return bool(self._s.done)

    def reset(self):
        lib().oracle_env_reset(ctypes.byref(self._s))
        self.history = []
        return self.observation()

    def observation(self):
        n = self.board_size
        out = np.empty((17, n, n), dtype=np.int8)
        lib().oracle_env_observation(ctypes.byref(self._s), out.ctypes.data)
        return out

    def step(self, action):
        reward, done = ctypes.c_double(0.0), ctypes.c_int(0)
        rc = lib().oracle_env_step(ctypes.byref(self._s), int(action), ctypes.byref(reward), ctypes.byref(done))
        if rc == -1:
            raise RuntimeError("Game is over, call reset before using step method.")
        if rc == -2:
            raise ValueError(f"Invalid action. The action {action} is out of bound.")
        if rc == -3:
            raise ValueError(f"Illegal action {action}.")
        if action != self.resign_move:
            self.history.append(int(action))
        return self.observation(), float(reward.value), bool(done.value), {}

    def record(self, reward=0.0):
        buf = (ctypes.c_uint8 * 1024)()
        k = lib().oracle_env_record(ctypes.byref(self._s), float(reward), buf)
        return bytes(buf[:k])

    def get_player_name_by_id(self, pid):
        return "B" if pid == self.black_player else "W" if pid == self.white_player else None

    def __deepcopy__(self, memo):
        new = object.__new__(type(self))
        new.__dict__.update(self.__dict__)
        new._s = _EnvStruct()
        ctypes.memmove(ctypes.byref(new._s), ctypes.byref(self._s), ctypes.sizeof(_EnvStruct))
        new.history = list(self.history)
        return new


class OracleGoEnv(OracleEnv):
    """GoEnv (go.py:19-86): colours +1/-1, pass = N*N, resign = -1."""

    kind = 0
    has_pass_move = True
    has_resign_move = True
    black_player, white_player = 1, -1
    legal_dtype = np.int64

    def __init__(self, board_size=9, komi=7.5, max_steps=0):
        super().__init__(board_size, komi=komi, max_steps=max_steps)

    @property
    def ko(self):
        return int(self._s.ko)

    @property
    def caps(self):
        return int(self._s.caps[0]), int(self._s.caps[1])

    def area_score(self):
        b, w = ctypes.c_int(0), ctypes.c_int(0)
        brd = np.ascontiguousarray(self.board)
        lib().oracle_go_area_score(brd.ctypes.data, self.board_size, ctypes.byref(b), ctypes.byref(w))
        return b.value, w.value

    def position_result(self):
        # go_engine.py:527-534 Position.result_string
        s = lib().oracle_go_score(ctypes.byref(self._s))
        return "B+%.1f" % s if s > 0 else "W+%.1f" % abs(s) if s < 0 else "DRAW"

    def get_result_string(self):
        # go.py:194-200
        if self._s.last_move == -1:
            return "B+R" if self.winner == self.black_player else "W+R"
        return self.position_result()


class OracleGomokuEnv(OracleEnv):
    """GomokuEnv (gomoku.py:17-43): colours 1/2, no pass, no resign."""

    kind = 1

    def __init__(self, board_size=15, num_to_win=5):
        super().__init__(board_size, num_to_win=num_to_win)

    def get_result_string(self):
        if not self.is_game_over():
            return ""
        return "B+1.0" if self.winner == 1 else "W+1.0" if self.winner == 2 else "DRAW"


def replay_digests(env, moves):
    """Replay a move list through the C oracle; returns (played, state_digest16, obs_digest16)."""
    import hashlib

    n = env.board_size
    moves = np.ascontiguousarray(moves, dtype=np.int32)
    rec = np.empty((len(moves) + 1) * (2 * n * n + 16), dtype=np.uint8)
    obs = np.empty((len(moves) + 1) * 17 * n * n, dtype=np.int8)
    rl, ol = ctypes.c_int64(0), ctypes.c_int64(0)
    played = lib().oracle_env_replay(ctypes.byref(env._s), moves.ctypes.data, len(moves), rec.ctypes.data, ctypes.byref(rl),
                                     obs.ctypes.data, ctypes.byref(ol))
    ds = hashlib.sha256(rec[: rl.value].tobytes()).digest()[:16]
    do = hashlib.sha256(obs[: ol.value].tobytes()).digest()[:16]
    return played, ds, do
