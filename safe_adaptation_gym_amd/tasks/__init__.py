from safe_adaptation_gym_amd.tasks.go_to_goal import (CatchGoal, GoToGoal, GoToGoalDamping,
                                                      GoToGoalMotor, GoToGoalScarce, Unsupervised)
from safe_adaptation_gym_amd.tasks.press_buttons import Collect, PressButtons, PressButtonsScarce
from safe_adaptation_gym_amd.tasks.push_box import (DribbleBall, HaulBox, PushBox, PushBoxScarce,
                                                    RollRod)
from safe_adaptation_gym_amd.tasks.task import Task

__all__ = [
    'GoToGoal', 'PushBox', 'PressButtons', 'RollRod', 'DribbleBall', 'Collect', 'HaulBox',
    'CatchGoal', 'Unsupervised', 'GoToGoalScarce', 'PressButtonsScarce', 'PushBoxScarce',
    'GoToGoalDamping', 'GoToGoalMotor'
]
