class DataLoader:  # type name only (reference model/Embedding.py:9); the hot path never iterates one
    pass
