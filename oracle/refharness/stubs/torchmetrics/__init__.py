class Dice:  # name only (fine-tune path, out of scope)
    def __init__(self, *a, **k):
        raise NotImplementedError
