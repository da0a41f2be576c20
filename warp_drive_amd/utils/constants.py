"""Reserved array names (reference warp_drive/utils/constants.py:11-21)."""


class Constants:
    OBSERVATIONS = "observations"
    ACTIONS = "sampled_actions"
    REWARDS = "rewards"
    DONE_FLAGS = "done_flags"
    PROCESSED_OBSERVATIONS = "processed_observations"
    ACTION_MASK = "action_mask"
