"""Online (on-policy) history is not part of the MI355X hot path (SURVEY.md
section 8 marks it 'plumbing only'); the key stays registered so configs that
name it fail with a clear message instead of a KeyError."""


class OnlineHistoryBuffer:
    def __init__(self, *a, **kw):
        raise NotImplementedError(
            "history mode 'online' (A2C/PPO) is outside the scope of rltime_amd; "
            "see DESIGN.md 'Out of scope'")
