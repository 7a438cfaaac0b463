from dataclasses import dataclass
from typing import Mapping, Optional, Tuple, Type

import numpy as np

from safe_adaptation_gym_amd.tasks.task import Task


@dataclass
class TaskSampler:
  """reference benchmark/task_sampler.py:10-19: first element of a permutation of the
  (name, class) items, instantiated."""
  rs: np.random.RandomState
  tasks: Mapping[str, Type[Task]]

  def sample(self) -> Optional[Tuple[str, Task]]:
    if len(self.tasks) == 0:
      return None
    name, task = self.rs.permutation(list(self.tasks.items()))[0]
    return name, task()
