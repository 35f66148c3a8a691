#include "tf_mock.h"
