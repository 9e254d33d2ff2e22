import enum


class RunningStage(enum.Enum):
    SANITY_CHECKING = "sanity_check"
    TRAINING = "train"
    VALIDATING = "validate"
