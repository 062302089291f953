"""Option-file reader with the reference's keys (reference src/main.cpp:646-683, README.md:208-250).

One `-key value` per line; unknown lines are ignored.  (Blank lines crash the reference, src/main.cpp:654;
they are skipped here.)  Defaults are the reference's (src/main.cpp:74-75,103-115)."""
from ._abi import Options

_KEYS = {"-dispw": ("width", int), "-disph": ("height", int), "-hfov": ("hfov", float), "-spp": ("spp", int),
         "-bounce": ("max_bounce", int), "-spr": ("spr", int), "-eps": ("eps", float), "-reps": ("ray_eps", float),
         "-rbounce": ("max_refrac_bounce", int)}


def parse_options(text, base=None):
    """Returns (Options, extras) where extras holds keys that are host-only (-mpbs: display block size)."""
    opt = base or Options()
    extras = {"mpbs": 100}
    for line in text.splitlines():
        tok = line.split()
        if len(tok) < 2:
            continue
        if tok[0] in _KEYS:
            name, conv = _KEYS[tok[0]]
            setattr(opt, name, conv(float(tok[1])) if conv is int else conv(tok[1]))
        elif tok[0] == "-mpbs":
            extras["mpbs"] = int(float(tok[1]))
    return opt, extras
