/* drop-in name for scene sources written against the reference; see rt_scene_api.h */
#include "rt_scene_api.h"
