class ModelCheckpoint:
    def __init__(self, *a, **k):
        pass

    def on_train_batch_end(self, *a, **k):
        pass
