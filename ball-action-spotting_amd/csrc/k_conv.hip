#include "elem.h"
