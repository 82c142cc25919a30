class SummaryWriter:  # name only
    def __init__(self, *a, **k):
        pass
