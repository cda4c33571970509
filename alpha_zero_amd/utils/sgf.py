"""SGF records and checkpoint dictionaries in the reference's formats.

  * make_sgf: the text `alpha_zero/utils/sgf_wrapper.py:59-91` writes (header properties CA, AP, RU, PB, BR, PW, WR, KM, RE, DT,
    SZ, one node per move, a line break after every 10th move, `]` escaped in comments); pinned byte-for-byte against the
    reference's output in tests/golden/sgf_records.json.
  * save_checkpoint / load_checkpoint: the dictionary of `core/pipeline.py:597-606` ('network', 'optimizer', 'lr_scheduler',
    'training_steps'), so runs can resume from / hand over to the reference (AlphaZeroNet here is state_dict-compatible)."""
import itertools
import time

from ..envs.coords import CoordsConvertor

_HEADER = ("CA[UTF-8]", "AP[AlphaZeroMini_sgfgenerator]", "RU[{ruleset}]", "PB[{black_name}]", "BR[{black_rank}]", "PW[{white_name}]",
           "WR[{white_rank}]", "KM[{komi}]", "RE[{result}]", "DT[{date}]", "SZ[{boardsize}]")


def get_time_stamp(file_name=False):
    """utils/util.py:15-20"""
    return time.strftime("%Y%m%d_%H%M%S" if file_name else "%Y-%m-%d %H:%M:%S", time.localtime())


def _move_node(cc, player_move, comment):
    if player_move.color not in ("B", "W"):
        raise ValueError("Can't translate color %s to sgf" % player_move.color)
    node = ";{}[{}]".format(player_move.color, cc.to_sgf(cc.from_flat(player_move.move)))
    if comment is not None:
        node += "C[{}]".format(comment.replace("]", r"\]"))
    return node


def make_sgf(board_size, move_history, result_string, ruleset="Chinese", komi=7.5, white_name="AlphaZeroMini", white_rank="",
             black_name="AlphaZeroMini", black_rank="", date="", comments=()):
    """move_history: iterable of (color 'B'/'W', flat move; board_size**2 = pass); comments are zipped with the moves."""
    cc = CoordsConvertor(board_size)
    nodes = [_move_node(cc, m, c) for m, c in itertools.zip_longest(move_history, comments)]
    body = "".join(n + "\n" if (i + 1) % 10 == 0 else n for i, n in enumerate(nodes))
    head = "\n".join(_HEADER).format(ruleset=ruleset, black_name=black_name, black_rank=black_rank, white_name=white_name, white_rank=white_rank,
                                    komi=komi, result=result_string, date=date, boardsize=board_size)
    return "(;\n" + head + "\n\n" + body + ")"


def save_checkpoint(path, network, optimizer=None, lr_scheduler=None, training_steps=0):
    import torch

    state = {"network": network.state_dict(), "training_steps": training_steps}
    if optimizer is not None:
        state["optimizer"] = optimizer.state_dict()
    if lr_scheduler is not None:
        state["lr_scheduler"] = lr_scheduler.state_dict()
    torch.save(state, path)


def load_checkpoint(path, network, optimizer=None, lr_scheduler=None, map_location="cpu"):
    """Returns training_steps (pipeline.py:243-246, :447-453 read the same keys)."""
    import torch

    state = torch.load(path, map_location=map_location)
    network.load_state_dict(state["network"])
    if optimizer is not None and "optimizer" in state:
        optimizer.load_state_dict(state["optimizer"])
    if lr_scheduler is not None and "lr_scheduler" in state:
        lr_scheduler.load_state_dict(state["lr_scheduler"])
    return state.get("training_steps", 0)
